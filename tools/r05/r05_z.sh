#!/bin/bash
# round 5, call z: X groups on a second stream (the iteration's priority): parity with a reduced grid, then the workgroup count at
# 8 and 4 ranks of 512^3, 384^3 on 8, config 5 on 8 / 4 ranks
set -o pipefail
EC3D_XASYNC_WGS=8 timeout -k 10 900 python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rank_loopback.py -x -q > gpurun_out/r05_z_slabs.log 2>&1
rc=$?; tail -n 5 gpurun_out/r05_z_slabs.log; [ $rc -eq 0 ] || exit $rc
EC3D_XASYNC=2 EC3D_XASYNC_WGS=8 timeout -k 10 900 python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_parity.py -x -q > gpurun_out/r05_z_forced.log 2>&1
rc=$?; tail -n 5 gpurun_out/r05_z_forced.log; [ $rc -eq 0 ] || exit $rc
for v in "EC3D_XASYNC=0" "EC3D_XASYNC_WGS=64" "EC3D_XASYNC_WGS=128" "EC3D_XASYNC_WGS=256" "EC3D_XASYNC_WGS=512" "EC3D_XASYNC=0"; do
  echo "== $v" >> gpurun_out/r05_z.log
  env $v REHEARSE_ONLY="512,512,8,4;384,384,8,4" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_z.log || exit 1
done
for v in "EC3D_XASYNC=0" "EC3D_XASYNC_WGS=128" "EC3D_XASYNC_WGS=256" "EC3D_XASYNC=0"; do
  echo "== $v" >> gpurun_out/r05_z.log
  env $v REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_z.log || exit 1
done
cut -c1-300 gpurun_out/r05_z.log
