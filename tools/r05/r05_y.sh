#!/bin/bash
# round 5, call y: X groups on a second stream at 8 ranks: stream priority and workgroup count
for v in "EC3D_XASYNC=0" "EC3D_XASYNC=1 EC3D_XASYNC_PRIO=0" "EC3D_XASYNC=1 EC3D_XASYNC_PRIO=0 EC3D_XASYNC_WGS=256" "EC3D_XASYNC=1 EC3D_XASYNC_PRIO=0 EC3D_XASYNC_WGS=64" "EC3D_XASYNC=1 EC3D_XASYNC_WGS=64" "EC3D_XASYNC=0" "EC3D_XASYNC=1 EC3D_XASYNC_PRIO=0"; do
  echo "== $v" >> gpurun_out/r05_y.log
  env $v REHEARSE_ONLY="512,512,8,4;512,512,2,1" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_y.log || exit 1
done
cut -c1-330 gpurun_out/r05_y.log
