#!/bin/bash
# round 5, call ah: A-V slabs on plan 5: does it run, does it agree with plan 2, what does a rank gain
set -o pipefail
timeout -k 10 300 python3 tools/av_plan5_check.py 64 32 48 2 > gpurun_out/r05_ah_check.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r05_ah_check.log | tail -n 12; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 tools/av_plan5_check.py 128 64 64 4 >> gpurun_out/r05_ah_check.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r05_ah_check.log | tail -n 4; [ $rc -eq 0 ] || exit $rc
for v in 2 5 2 5; do
  echo "== EC3D_SLAB_PLAN=$v" >> gpurun_out/r05_ah.log
  EC3D_SLAB_PLAN=$v REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_ah.log || exit 1
done
cut -c1-300 gpurun_out/r05_ah.log
