#!/bin/bash
# round 6, call r (third run: mode 5): patch_pair request modes per operand vector (tools/ab_build.sh variants), 512^3, same box, interleaved
out=$(pwd)/gpurun_out/r06r; mkdir -p $out
for rep in 1 2 3; do
  for v in base L2P2 F4P3 F4P5 F5P5 L5F4P3 L3F4P3; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 tools/ab_perf.py cube512 $v 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
