#!/bin/bash
# round 6, call s: K5-in-K1's AP.R0 a step later (EC3D_K51_R0_DEFER); X groups beside the iteration with the new kernels
out=$(pwd)/gpurun_out/r06s; mkdir -p $out
for rep in 1 2 3; do
  for v in base new newR0; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 tools/ab_perf.py cube512 $v 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
  EC3D_XASYNC=2 EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_new.so timeout -k 10 300 python3 tools/ab_perf.py cube512 new_xasync2 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  EC3D_XASYNC=2 EC3D_XASYNC_WGS=0 EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_new.so timeout -k 10 300 python3 tools/ab_perf.py cube512 new_xasync2_wgs0 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
