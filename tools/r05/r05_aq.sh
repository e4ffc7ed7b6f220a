#!/bin/bash
# round 5, call aq: full-size multi-GPU forms of bench.py on one card with the final defaults (verification included)
set -o pipefail
timeout -k 10 500 python bench.py --gpus 2 --devices 0,0 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/r05_aq_lib2.json 2> gpurun_out/r05_aq.err || { tail -n 5 gpurun_out/r05_aq.err; exit 1; }
timeout -k 10 500 python bench.py --rehearse 4,8 --steps 100 --warmup 5 --no-cpu-baseline > gpurun_out/r05_aq_reh.json 2>> gpurun_out/r05_aq.err || { tail -n 5 gpurun_out/r05_aq.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/r05_aq_lib2.json", "gpurun_out/r05_aq_reh.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("parallelism", "")[-120:], "|", d.get("verified"), "|", d.get("host"))
PY
