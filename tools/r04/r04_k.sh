#!/bin/bash
out=gpurun_out/r04k; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 100 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in lim av3; do
  run default $wl A=1
  for k in 0 1 8 9 11 41 43; do run keep$k $wl EC3D_KEEP=$k; done
done
done
# config 5 with the mid-size plans of the vector kernels instead of the big ones
for rep in 1 2; do
  run vec768_xcd_d1 lim EC3D_NBLK_K2=768 EC3D_NBLK_K4=768 EC3D_NBLK_K5=768 EC3D_XCD_MAP=1 EC3D_VEC_DEPTH=1
  run vec768_xcd_d1_keep9 lim EC3D_NBLK_K2=768 EC3D_NBLK_K4=768 EC3D_NBLK_K5=768 EC3D_XCD_MAP=1 EC3D_VEC_DEPTH=1 EC3D_KEEP=9
  run k4_512 lim EC3D_NBLK_K4=512
  run k4_512_keep9 lim EC3D_NBLK_K4=512 EC3D_KEEP=9
done
cat $out/ab.log
