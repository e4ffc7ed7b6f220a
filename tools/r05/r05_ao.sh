#!/bin/bash
# round 5, call ao: bench.py's own verification of the multi-GPU forms at scale on the final defaults (A-V plan 5 + U planes left at home; cube plan 5 / 1)
set -o pipefail
timeout -k 10 500 python bench.py --gpus 4 --devices 0,0,0,0 --workload av --refine 3 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ao_av4.json 2> gpurun_out/r05_ao.err || { tail -n 5 gpurun_out/r05_ao.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 2 --devices 0,0 --workload av --refine 2 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ao_av2.json 2>> gpurun_out/r05_ao.err || { tail -n 5 gpurun_out/r05_ao.err; exit 1; }
timeout -k 10 500 python bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0 --grid 256 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r05_ao_cube8.json 2>> gpurun_out/r05_ao.err || { tail -n 5 gpurun_out/r05_ao.err; exit 1; }
python3 - <<'PY'
import json
for f in ("gpurun_out/r05_ao_av4.json", "gpurun_out/r05_ao_av2.json", "gpurun_out/r05_ao_cube8.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["config"].get("parallelism", "")[-140:], "|", d.get("verified"))
PY
