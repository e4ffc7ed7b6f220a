#!/bin/bash
# round 5, call x: the groups of X updates on a stream of their own -- parity first (slabs: default; single handles: forced),
# then the rank rehearsal with and without
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rank_loopback.py tests/test_gpu_rccl_rank.py tests/test_gpu_multi.py -x -q > gpurun_out/r05_x_slabs.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_x_slabs.log; [ $rc -eq 0 ] || exit $rc
EC3D_XASYNC=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/r05_x_single.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_x_single.log; [ $rc -eq 0 ] || exit $rc
for v in "EC3D_XASYNC=0" "EC3D_XASYNC=1" "EC3D_XASYNC=1 EC3D_XASYNC_WGS=256" "EC3D_XASYNC=1 EC3D_XASYNC_WGS=512" "EC3D_XASYNC=0" "EC3D_XASYNC=1"; do
  echo "== $v" >> gpurun_out/r05_x_rehearsal.log
  env $v REHEARSE_ONLY="512,512,8,4;512,512,4,2;512,512,2,1" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 >> gpurun_out/r05_x_rehearsal.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/r05_x_rehearsal.log | cut -c1-330
