#!/bin/bash
# round 5, call i: the in-kernel collapse with all its loads requested at once, against the collapse launches
set -o pipefail
mkdir -p gpurun_out/r05i
python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_multi.py -x -q > gpurun_out/r05i/parity.log 2>&1; echo "parity rc=$?" | tee -a gpurun_out/r05i/summary.log
for cfg in "1 0" "1 1" "1 0" "1 1" "2 1"; do
  set -- $cfg
  echo "== EC3D_SLAB_PLAN=$1 EC3D_FOLD=$2" >> gpurun_out/r05i/fold.log
  EC3D_SLAB_PLAN=$1 EC3D_FOLD=$2 REHEARSE_ONLY="512,512,8,4;512,512,4,2;256,256,8,3;64,128,8,3" timeout -k 10 200 python tools/rank_rehearsal.py 300 >> gpurun_out/r05i/fold.log 2>&1
done
tail -n 3 gpurun_out/r05i/parity.log
grep -v "version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05i/fold.log
