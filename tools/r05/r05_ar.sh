#!/bin/bash
# round 5, call ar: the send / recv kernel made resident BEFORE the iteration's next launch (the compute stream waits for the side stream to reach it)
for v in "A=0" "EC3D_MULTI_EXCHANGE_FIRST=1" "A=0" "EC3D_MULTI_EXCHANGE_FIRST=1"; do
  echo "== $v" >> gpurun_out/r05_ar.log
  env $v REHEARSE_ONLY="512,512,8,4;384,384,8,4;512,512,4,2" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_ar.log || exit 1
  env $v REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_ar.log || exit 1
done
cut -c1-250 gpurun_out/r05_ar.log
