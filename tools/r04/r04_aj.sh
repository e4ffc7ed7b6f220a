#!/bin/bash
# round 4, call aj: is the deferred X update with one tile in flight without X and two in the applying launch a gain at the
# sizes between the caches and 32 Mi rows?  three runs each, alternating
out=gpurun_out/r04aj; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2 3; do
for wl in hole lim av3 cube256 box:256x256x80 av2; do
  run classic $wl EC3D_XDEFER=1
  run d4_off1_on2 $wl EC3D_XDEFER=4 EC3D_XD_OFF_DEPTH=1 EC3D_XD_ON_DEPTH=2
done
done
sort -k1,2 -s $out/ab.log
