#!/bin/bash
set -e
python -m pytest tests/test_gpu_parity.py -x -q -k "two_dimensional" > gpurun_out/ab_pytest.log 2>&1 || { tail -20 gpurun_out/ab_pytest.log; exit 1; }
tail -2 gpurun_out/ab_pytest.log
for rep in 1 2; do
  EC3D_LIB=tools/ab/libec3d_hip_r0late.so python3 tools/ab_perf.py cube512 r0_late
  python3 tools/ab_perf.py cube512 r0_early
done
