#!/bin/bash
# round 4, call w: tiles in flight of the two kinds of deferred-X K4 launches, 512^3
out=gpurun_out/r04w; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
run classic cube512 EC3D_XDEFER=1
for d in 2 4; do
  for off in 1 2 4; do for on in 1 2; do
    run d${d}_off${off}_on${on} cube512 EC3D_XDEFER=$d EC3D_XD_OFF_DEPTH=$off EC3D_XD_ON_DEPTH=$on
  done; done
done
run classic cube512 EC3D_XDEFER=1
cat $out/ab.log
