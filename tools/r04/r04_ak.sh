#!/bin/bash
# round 4, call ak: group length of the deferred X update at the mid sizes (rings of 2, 3, 4 buffers beside the Infinity Cache)
out=gpurun_out/r04ak; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in hole lim av3 cube256; do
  for d in 1 2 3 4; do run every_$d $wl EC3D_XDEFER=$d; done
done
done
sort -k1,2 -s $out/ab.log
