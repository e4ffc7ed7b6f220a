#!/bin/bash
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
  EC3D_KEEP_S=1 python3 tools/ab_perf.py cube512 S_stored
  python3 tools/ab_perf.py cube512 S_formed_again
done
for cfg in "256 0 2" "256 0 4" "512 0 2" "256 0 1" "768 0 2" "256 1 2"; do set -- $cfg; EC3D_NBLK_K4=$1 EC3D_MAP_K4=$2 EC3D_DEPTH_K4=$3 python3 tools/ab_perf.py cube512 "k4 nblk=$1 map=$2 depth=$3"; done
