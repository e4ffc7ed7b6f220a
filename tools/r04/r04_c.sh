#!/bin/bash
out=gpurun_out/r04c; mkdir -p $out
for wl in lim av3; do
  for nb in 1024 1280 1536 2048; do
    EC3D_NBLK_SPMV=$nb EC3D_FUSE23=2 EC3D_FUSE51=2 timeout -k 10 200 python3 tools/ab_perf.py $wl fused_nb$nb >> $out/ab.log 2>> $out/ab.err
    EC3D_NBLK_SPMV=$nb EC3D_FUSE23=0 EC3D_FUSE51=0 timeout -k 10 200 python3 tools/ab_perf.py $wl patch_nb$nb >> $out/ab.log 2>> $out/ab.err
    EC3D_NBLK_SPMV=$nb EC3D_FUSE23=0 EC3D_FUSE51=0 EC3D_SAV_PATCH=0 timeout -k 10 200 python3 tools/ab_perf.py $wl linear_nb$nb >> $out/ab.log 2>> $out/ab.err
  done
done
cat $out/ab.log
