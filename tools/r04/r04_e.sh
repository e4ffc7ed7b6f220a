#!/bin/bash
out=gpurun_out/r04e; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 100 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in av1 lim0; do
  run base $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2
  run depth2 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2 EC3D_VEC_DEPTH=2
  run depth4 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2 EC3D_VEC_DEPTH=4
  run vec1536 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2 EC3D_NBLK_K2=1536 EC3D_NBLK_K4=1536 EC3D_NBLK_K5=1536
  run vec2048plain $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2 EC3D_NBLK_K2=2048 EC3D_NBLK_K4=2048 EC3D_NBLK_K5=2048 EC3D_XCD_MAP=0
  run vec1536plain $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2 EC3D_NBLK_K2=1536 EC3D_NBLK_K4=1536 EC3D_NBLK_K5=1536 EC3D_XCD_MAP=0
  run patch_pps2 $wl EC3D_SAV_PATCH=2 EC3D_MIN_PPS=2 EC3D_FUSE23=0 EC3D_FUSE51=0
  run fused_pps2 $wl EC3D_SAV_PATCH=2 EC3D_MIN_PPS=2 EC3D_FUSE23=2 EC3D_FUSE51=2
  run fused_pps1 $wl EC3D_SAV_PATCH=2 EC3D_MIN_PPS=1 EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_NBLK_SPMV=2048
done
done
for wl in cube128 box:256x256x64 av2; do
  run pps8 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=8
  run pps4 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=4
  run pps2 $wl EC3D_SAV_PATCH=0 EC3D_MIN_PPS=2
done
for wl in box:512x512x256 box:512x512x384; do
  run fuse0 $wl EC3D_FUSE23=0 EC3D_FUSE51=0
  run fuse2 $wl EC3D_FUSE23=2 EC3D_FUSE51=2
done
cat $out/ab.log
