#!/bin/bash
# round 6, call u: K5-in-K1's R0 pair behind the arrival of the plane above (with the rim row), AP.R0 before the stores
out=$(pwd)/gpurun_out/r06u; mkdir -p $out
for rep in 1 2 3; do
  for v in new new2 r0late dotfirst late; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 tools/ab_perf.py cube512 $v 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
