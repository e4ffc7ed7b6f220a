#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
for wl in av3 cube256 cube512; do
timeout -k 10 300 python tools/vec_sweep.py $wl ";NT=0;NT=1;" > $out/nt_$wl.log 2>> $out/sweep.err
cat $out/nt_$wl.log
done
