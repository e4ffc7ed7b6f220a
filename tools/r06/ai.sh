#!/bin/bash
# round 6, call ai: where the three-launch iteration starts to pay with this round's kernels (five launches by default below 32 Mi rows)
out=$(pwd)/gpurun_out/r06ai; mkdir -p $out
for rep in 1 2; do
for g in 256x256x256 512x512x64 512x512x72 384x384x128; do
  timeout -k 10 300 python3 tools/ab_perf.py box:$g five_launches 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_K4S=2 EC3D_XDEFER=4 timeout -k 10 300 python3 tools/ab_perf.py box:$g three_launches_forced 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
done
done
