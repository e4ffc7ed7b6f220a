#!/bin/bash
# round 6, call al: 256^3 (BASELINE config 2): one fusion at a time with this round's z-march step
out=$(pwd)/gpurun_out/r06al; mkdir -p $out
for rep in 1 2; do
  timeout -k 10 300 python3 tools/ab_perf.py cube256 five_launches 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_FUSE23=2 timeout -k 10 300 python3 tools/ab_perf.py cube256 K2_in_K3_only 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_FUSE51=2 timeout -k 10 300 python3 tools/ab_perf.py cube256 K5_in_K1_only 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_K4S=0 timeout -k 10 300 python3 tools/ab_perf.py cube256 both_K4_classic 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_K4S=2 timeout -k 10 300 python3 tools/ab_perf.py cube256 three_launches 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
done
