#!/bin/bash
out=gpurun_out/r04m; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 100 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for wl in lim hole; do
  run default $wl A=1
  for nb in 1024 1280 1536 2048 3072; do run spmv$nb $wl EC3D_NBLK_SPMV=$nb; done
  for nb in 512 1024 1536; do run vec$nb $wl EC3D_NBLK_K2=$nb EC3D_NBLK_K4=$nb EC3D_NBLK_K5=$nb; done
  run k4_512 $wl EC3D_NBLK_K4=512
  run k4_1024 $wl EC3D_NBLK_K4=1024
  run depth2 $wl EC3D_VEC_DEPTH=2
  run plainmap $wl EC3D_XCD_MAP=0
  run default2 $wl A=1
done
cat $out/ab.log
