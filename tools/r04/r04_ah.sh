#!/bin/bash
# round 4, call ah: the deferred X update at the A-V configs' sizes with other tiles in flight than the big-grid defaults
out=gpurun_out/r04ah; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for wl in av3 lim; do
  run classic $wl EC3D_XDEFER=1
  for off in 1 2 4; do for on in 1 2; do
    run d4_off${off}_on${on} $wl EC3D_XDEFER=4 EC3D_XD_OFF_DEPTH=$off EC3D_XD_ON_DEPTH=$on
  done; done
  run classic $wl EC3D_XDEFER=1
done
cat $out/ab.log
