#!/bin/bash
# round 6, call g: the structured kernels WITHOUT a conductor against the cube's (rows per second of the bare band part)
out=$(pwd)/gpurun_out/r06g; mkdir -p $out
run() { local name=$1; shift; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; tail -n 1 $out/perf.log | cut -c1-420; }
run air_off AIR=1 EC3D_SAV_IL=0
run air_il AIR=1 EC3D_SAV_IL=1
run air_il1024 AIR=1 EC3D_SAV_IL=1 EC3D_NBLK_SPMV=1024
run air_off_1536 AIR=1 EC3D_SAV_IL=0 EC3D_NBLK_SPMV=1536
EC3D_PATCH=0 python tools/cube_perf.py cube_linear 256 | tee -a $out/perf.log
EC3D_PATCH=1 python tools/cube_perf.py cube_patch 256 | tee -a $out/perf.log
EC3D_PATCH=0 EC3D_NBLK_SPMV=1024 python tools/cube_perf.py cube_linear_1024 256 | tee -a $out/perf.log
EC3D_PATCH=0 EC3D_NBLK_SPMV=1536 python tools/cube_perf.py cube_linear_1536 256 | tee -a $out/perf.log
