#!/bin/bash
out=gpurun_out/r04d; mkdir -p $out
export EC3D_SAV_PATCH=0
for rep in 1 2; do
for wl in av1; do
  EC3D_LIB=tools/ab/libec3d_hip_r03.so timeout -k 10 100 python3 tools/ab_perf.py $wl r03 >> $out/ab.log 2>> $out/ab.err
  timeout -k 10 100 python3 tools/ab_perf.py $wl default >> $out/ab.log 2>> $out/ab.err
  EC3D_ZMARCH=0 timeout -k 10 100 python3 tools/ab_perf.py $wl nozmarch >> $out/ab.log 2>> $out/ab.err
  for pps in 1 2 4; do for nb in 768 1536 3072; do
    EC3D_MIN_PPS=$pps EC3D_NBLK_SPMV=$nb timeout -k 10 100 python3 tools/ab_perf.py $wl pps${pps}_nb$nb >> $out/ab.log 2>> $out/ab.err
  done; done
done
done
# the prologue change at the sizes where it should not matter
for wl in av3 cube256; do
  EC3D_LIB=tools/ab/libec3d_hip_r03.so timeout -k 10 100 python3 tools/ab_perf.py $wl r03 >> $out/ab.log 2>> $out/ab.err
  timeout -k 10 100 python3 tools/ab_perf.py $wl r04 >> $out/ab.log 2>> $out/ab.err
done
# fusion threshold on the cube
for wl in box:512x512x128 cube384; do
for f in 0 2; do EC3D_FUSE23=$f EC3D_FUSE51=$f timeout -k 10 100 python3 tools/ab_perf.py $wl fuse$f >> $out/ab.log 2>> $out/ab.err; done
done
cat $out/ab.log
