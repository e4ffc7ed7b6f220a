#!/bin/bash
# scratch A/B (one gpurun call): 2-D tiles, rim rows one plane ahead
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats_dist.py tests/test_gpu_multi.py -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
  EC3D_LIB=tools/ab/libec3d_hip_rim0.so python3 tools/ab_perf.py cube512 rim_now
  python3 tools/ab_perf.py cube512 rim_ahead
done
EC3D_LIB=tools/ab/libec3d_hip_rim0.so EC3D_FUSE23=0 EC3D_FUSE51=0 python3 tools/ab_perf.py cube512 rim_now_5launch
EC3D_FUSE23=0 EC3D_FUSE51=0 python3 tools/ab_perf.py cube512 rim_ahead_5launch
EC3D_LIB=tools/ab/libec3d_hip_rim0.so python3 tools/ab_perf.py cube256 rim_now
python3 tools/ab_perf.py cube256 rim_ahead
