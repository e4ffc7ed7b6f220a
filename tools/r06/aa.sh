#!/bin/bash
# round 6, call aa: the X-group launch's grid (one process, placement fixed by the search; per-kernel K4 = light K4 + group / 4)
out=$(pwd)/gpurun_out/r06aa; mkdir -p $out
for rep in 1 2; do
for w in 0 256 384 512 768; do
  EC3D_XGROUP_WGS=$w timeout -k 10 300 python3 tools/ab_perf.py cube512 xgroup_wgs=$w 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
done
