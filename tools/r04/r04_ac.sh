#!/bin/bash
# round 4, call ac: workgroups of the SpMV-form kernels (K23, K4s, K51) at 512^3 now that they move fewer bytes per row
out=gpurun_out/r04ac; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
  run nblk1024_default cube512 A=1
  for nb in 512 1536 2048 3072; do run nblk$nb cube512 EC3D_NBLK_SPMV=$nb; done
done
cat $out/ab.log
