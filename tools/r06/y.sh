#!/bin/bash
# round 6, call y: physically contiguous memory for vectors and rings (hipExtMallocWithFlags, hipDeviceMallocContiguous)?
out=$(pwd)/gpurun_out/r06y; mkdir -p $out
EC3D_PLACE_VERBOSE=1 EC3D_VEC_CONTIG=1 EC3D_PLACE_VEC=0 timeout -k 10 400 python tools/vec_place_probe.py 4 > $out/contig.log 2>&1 || { tail $out/contig.log; exit 1; }
grep -v amdgpu.ids $out/contig.log
for i in 1 2 3; do
  for cg in 0 1; do
    EC3D_VEC_CONTIG=$cg EC3D_PLACE_VEC=0 EC3D_PLACE_VERBOSE=1 timeout -k 10 300 python3 tools/ab_perf.py cube512 contig=$cg 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
grep -c refused $out/ab.err
# X groups as launches of their own on the iteration's OWN stream (light K4 every iteration + one streaming launch per group)
for i in 1 2; do
  EC3D_PLACE_VEC=4 timeout -k 10 300 python3 tools/ab_perf.py cube512 fused_x 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  EC3D_PLACE_VEC=4 EC3D_XASYNC=2 EC3D_XASYNC_SAME=1 EC3D_XASYNC_WGS=0 timeout -k 10 300 python3 tools/ab_perf.py cube512 xgroup_same_stream 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  EC3D_PLACE_VEC=4 EC3D_XASYNC=2 EC3D_XASYNC_SAME=1 EC3D_XASYNC_WGS=1024 timeout -k 10 300 python3 tools/ab_perf.py cube512 xgroup_same_stream_1024 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
