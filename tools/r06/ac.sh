#!/bin/bash
# round 6, call ac: nontemporal loads for the plane above of the fused operand vectors (rim row and edge cells stay plain)
out=$(pwd)/gpurun_out/r06ac; mkdir -p $out
for rep in 1 2 3; do
  for v in cur ntmain; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 tools/ab_perf.py cube512 $v 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
