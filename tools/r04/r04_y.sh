#!/bin/bash
# round 4, call y: is the run-to-run spread of the 512^3 iteration (K51 1165 / 1230 us) a matter of how far apart the
# vectors lie?  EC3D_STAGGER=k puts k x 256 B more between consecutive vectors; five processes each
out=gpurun_out/r04y; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2 3 4 5; do
  for k in 0 1 17 273; do run stagger$k cube512 EC3D_STAGGER=$k; done
done
sort -k2,2 -s $out/ab.log
