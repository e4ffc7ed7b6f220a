#!/bin/bash
# round 6, call af: interleaved A-V march -- the step's +-sdx / edge requests only behind the arrival of the plane above
out=$(pwd)/gpurun_out/r06af; mkdir -p $out
for i in 1 2 3; do
  for v in il0 ildrain; do
  EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_$v.so timeout -k 10 400 python3 tools/av256_perf.py $v 2>> $out/av.err | tail -n 1 | cut -c1-20,150-420 | tee -a $out/av.log
  done
done
