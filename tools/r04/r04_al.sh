#!/bin/bash
# round 4, call al: which outputs stay cacheable (EC3D_KEEP: 1 AP, 2 S, 8 R, 32 P) now that X is updated every fourth
# iteration and P, S cycle through rings, at the mid sizes
out=gpurun_out/r04al; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for wl in hole lim av3 cube256; do
  run default $wl A=1
  for k in 0 1 9 11 41 43 33; do run keep$k $wl EC3D_KEEP=$k; done
  run default $wl A=1
done
cat $out/ab.log
