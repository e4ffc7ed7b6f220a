#!/bin/bash
# round 5, call av: X every fourth iteration on slabs below the 4.5 Mi-row rule
for v in "A=0" "EC3D_XDEFER=4" "A=0" "EC3D_XDEFER=4"; do
  echo "== $v" >> gpurun_out/r05_av.log
  env $v REHEARSE_ONLY="256,256,8,3;320,320,8,4" timeout -k 10 300 python3 tools/rank_rehearsal.py 300 2>&1 | grep "ms per iteration" >> gpurun_out/r05_av.log || exit 1
  env $v REHEARSE_AV=all timeout -k 10 300 python3 tools/rank_rehearsal.py 300 2>&1 | grep "hole.*rank . of" >> gpurun_out/r05_av.log || exit 1
done
cut -c1-230 gpurun_out/r05_av.log
