#!/bin/bash
# round 5, call p: A-V slabs of config 5 through the RCCL driver (rehearsal): producer-side split (plan 2) against no split (plan 0)
set -o pipefail
out=gpurun_out/r05p; mkdir -p $out
for plan in 2 0 2 0; do
  echo "== EC3D_SLAB_PLAN=$plan" >> $out/av_plans.log
  EC3D_SLAB_PLAN=$plan REHEARSE_AV=lim timeout -k 10 300 python tools/rank_rehearsal.py 300 2>&1 | grep "rank" >> $out/av_plans.log
done
cat $out/av_plans.log
