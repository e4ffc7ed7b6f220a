#!/bin/bash
# round 5, call ak: A-V ranks on plan 5 against plan 2: configs 5 and 3 on 2 / 4 / 8 ranks
for v in 2 5 2 5; do
  echo "== EC3D_SLAB_PLAN=$v" >> gpurun_out/r05_ak.log
  EC3D_SLAB_PLAN=$v REHEARSE_AV=all timeout -k 10 400 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_ak.log || exit 1
done
cut -c1-260 gpurun_out/r05_ak.log
