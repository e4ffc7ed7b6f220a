#!/bin/bash
# round 5, call e: which five-launch plan for the 16 Mi-row slabs of 512^3 on 8 GPUs (rehearsal of rank 4 over RCCL)
set -o pipefail
mkdir -p gpurun_out/r05e
for plan in 1 0 2 1 0 2; do
  echo "== EC3D_SLAB_PLAN=$plan" >> gpurun_out/r05e/plans.log
  EC3D_SLAB_PLAN=$plan REHEARSE_ONLY="512,512,8,4;256,256,8,3;384,128,8,3" timeout -k 10 200 python tools/rank_rehearsal.py 300 >> gpurun_out/r05e/plans.log 2>&1
done
grep -v "version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05e/plans.log
