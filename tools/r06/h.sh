#!/bin/bash
# round 6, call h: the interleaved march with a weight-balanced work list -- tests, then weights and grid sizes at 256^3
out=$(pwd)/gpurun_out/r06h; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_interleaved.py -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log; [ $rc -ne 0 ] && exit 1
run() { local name=$1; shift; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; tail -n 1 $out/perf.log | cut -c1-420; }
run w100 EC3D_IL_W=100
run w130 EC3D_IL_W=130
run w150 EC3D_IL_W=150
run w200 EC3D_IL_W=200
run w250 EC3D_IL_W=250
run w150_768 EC3D_IL_W=150 EC3D_NBLK_SPMV=768
run w150_1024 EC3D_IL_W=150 EC3D_NBLK_SPMV=1024
run w200_1024 EC3D_IL_W=200 EC3D_NBLK_SPMV=1024
run air_il AIR=1
run off EC3D_SAV_IL=0
