#!/bin/bash
# round 5, call ab: the driver's bench commands on the final tree (default and --steps 20)
set -o pipefail
timeout -k 10 500 python bench.py > gpurun_out/r05_ab_bench.json 2> gpurun_out/r05_ab_bench.err; rc=$?
tail -c 1500 gpurun_out/r05_ab_bench.json; [ $rc -eq 0 ] || { tail -n 20 gpurun_out/r05_ab_bench.err; exit $rc; }
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r05_ab_bench20.json 2>> gpurun_out/r05_ab_bench.err; rc=$?
python3 - <<'PY'
import json
for f in ("gpurun_out/r05_ab_bench.json", "gpurun_out/r05_ab_bench20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d.get("iter_dia", {}).get("frac"), d["cpu_baseline"]["value"])
PY
exit $rc
