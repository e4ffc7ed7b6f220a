#!/bin/bash
# round 4, call ab: with K4 as an SpMV kernel, from which size does the three-launch iteration pay?
out=gpurun_out/r04ab; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in cube256 box:512x512x96 box:512x512x128 cube384; do
  run default $wl A=1
  run fused_k4s_d4 $wl EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_K4S=2 EC3D_XDEFER=4
  run fused_k4s_d1 $wl EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_K4S=2 EC3D_XDEFER=1
done
done
cat $out/ab.log
