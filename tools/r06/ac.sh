#!/bin/bash
# round 6, call ac (second run): nontemporal loads for the plane above of the patch's two INNER rows only (nobody's rim rows)
out=$(pwd)/gpurun_out/r06ac; mkdir -p $out
for rep in 1 2 3; do
  for v in cur ntinner; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 tools/ab_perf.py cube512 $v 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
