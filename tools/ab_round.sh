#!/bin/bash
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_formats_dist.py tests/test_gpu_timeloop.py tests/test_gpu_two_process.py -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
 for wl in cube256 av3; do
  EC3D_KEEP=0 python3 tools/ab_perf.py $wl all_nontemporal
  python3 tools/ab_perf.py $wl policy
 done
done
EC3D_KEEP=0 python3 tools/ab_perf.py cube512 all_nontemporal
python3 tools/ab_perf.py cube512 policy
python3 bench.py --grid 256 --no-cpu-baseline --no-spmv-dia 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 256', d['value'], d['ms_per_step'])"
python3 bench.py --workload av --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench av', d['value'], d['ms_per_step'])"
