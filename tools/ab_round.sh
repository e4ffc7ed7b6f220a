#!/bin/bash
set -e
python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats_dist.py -x -q > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
  EC3D_LIB=tools/ab/libec3d_hip_r0late.so python3 tools/ab_perf.py cube512 before
  python3 tools/ab_perf.py cube512 raw_then_form
done
EC3D_LIB=tools/ab/libec3d_hip_r0late.so EC3D_FUSE23=0 EC3D_FUSE51=0 python3 tools/ab_perf.py cube512 before_5launch
EC3D_FUSE23=0 EC3D_FUSE51=0 python3 tools/ab_perf.py cube512 raw_5launch
EC3D_LIB=tools/ab/libec3d_hip_r0late.so python3 tools/ab_perf.py cube256 before
python3 tools/ab_perf.py cube256 raw_then_form
EC3D_FUSE23=2 EC3D_FUSE51=2 python3 tools/ab_perf.py cube256 raw_fused
