#!/bin/bash
# round 4, call af: the fused kernels compiled for 5 and 6 waves per SIMD (96 / 80 registers; K5-in-K1 spills 21 at 80) on
# grids of 1024 and 1536 workgroups, against the default (4 waves per SIMD, 1024)
out=gpurun_out/r04af; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
  run occ4_1024 cube512 A=1
  run occ5_1024 cube512 EC3D_LIB=tools/ab/libec3d_hip_occ5.so
  run occ6_1024 cube512 EC3D_LIB=tools/ab/libec3d_hip_occ6.so
  run occ6_1536 cube512 EC3D_LIB=tools/ab/libec3d_hip_occ6.so EC3D_NBLK_SPMV=1536
  run occ5_1536 cube512 EC3D_LIB=tools/ab/libec3d_hip_occ5.so EC3D_NBLK_SPMV=1536
done
cat $out/ab.log
