#!/bin/bash
# round 5, call h: producer-side split with window sweeps (plan 2 on the cube) against plans 0 / 1, fold on / off
set -o pipefail
mkdir -p gpurun_out/r05h
python -m pytest tests/test_gpu_slab_plans.py -x -q > gpurun_out/r05h/slab_plans.log 2>&1; echo "slab plans rc=$?" | tee -a gpurun_out/r05h/summary.log
for cfg in "1 0" "2 0" "0 0" "1 1" "2 1" "1 0" "2 0"; do
  set -- $cfg
  echo "== EC3D_SLAB_PLAN=$1 EC3D_FOLD=$2" >> gpurun_out/r05h/plans.log
  EC3D_SLAB_PLAN=$1 EC3D_FOLD=$2 REHEARSE_ONLY="512,512,8,4;256,256,8,3" timeout -k 10 200 python tools/rank_rehearsal.py 300 >> gpurun_out/r05h/plans.log 2>&1
done
tail -n 3 gpurun_out/r05h/slab_plans.log
grep -v "version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05h/plans.log
