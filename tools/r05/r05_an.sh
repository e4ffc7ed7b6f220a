#!/bin/bash
# round 5, call an: compute units kept free of the iteration's kernels for the send / recv kernel (EC3D_MULTI_CU_RESERVE)
for v in "A=0" "EC3D_MULTI_CU_RESERVE=8" "EC3D_MULTI_CU_RESERVE=16" "EC3D_MULTI_CU_RESERVE=32" "EC3D_MULTI_CU_RESERVE=8 EC3D_MULTI_CU_LAYOUT=1" "EC3D_MULTI_CU_RESERVE=16 EC3D_MULTI_CU_LAYOUT=1" "EC3D_MULTI_CU_RESERVE=32 EC3D_MULTI_CU_LAYOUT=1" "A=0"; do
  echo "== $v" >> gpurun_out/r05_an.log
  env $v REHEARSE_ONLY="512,512,8,4;384,384,8,4" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_an.log || exit 1
  env $v REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_an.log || exit 1
done
cut -c1-250 gpurun_out/r05_an.log
