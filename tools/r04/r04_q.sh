#!/bin/bash
out=gpurun_out/r04q; mkdir -p $out
timeout -k 10 200 python tools/dbg_quad.py 2>&1 | tee $out/dbg.log | tail -8
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 100 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in av3 hole lim; do
  run linear $wl A=1
  for nb in 768 1024 1536; do run quad_nb$nb $wl EC3D_SAV_QUAD=1 EC3D_NBLK_QUAD=$nb; done
done
done
cat $out/ab.log
