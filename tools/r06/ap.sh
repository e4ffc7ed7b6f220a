#!/bin/bash
# round 6, call ap: 640^3 and 768^3 -- the z-march grids the library picks, and others (EC3D_NBLK_SPMV)
out=$(pwd)/gpurun_out/r06ap; mkdir -p $out
for g in 640x640x640; do
  for nb in 0 800 1600 2400 3200 4000; do
    if [ $nb = 0 ]; then EC3D_PLACE_VEC=0 timeout -k 10 300 python3 tools/ab_perf.py box:$g default 2>> $out/ab.err | cut -c1-170 | tee -a $out/ab.log
    else EC3D_PLACE_VEC=0 EC3D_NBLK_SPMV=$nb timeout -k 10 300 python3 tools/ab_perf.py box:$g nblk_spmv=$nb 2>> $out/ab.err | cut -c1-170 | tee -a $out/ab.log; fi
  done
done
