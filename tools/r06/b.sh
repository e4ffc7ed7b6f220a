#!/bin/bash
# round 6, call b: the interleaved z-march of the structured kernels -- tests, then BASELINE config 3 at 256^3 with and without
out=gpurun_out/r06b; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_interleaved.py tests/test_gpu_rank_loopback.py tests/test_gpu_rccl_rank.py tests/test_gpu_multi.py -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; tail -n 6 $out/pytest.log; [ $rc -eq 124 ] && exit 1
run() { local name=$1; shift; echo "== $name"; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; local rc=$?; echo "rc=$rc"; tail -n 1 $out/perf.log; [ $rc -eq 124 ] && exit 1; return 0; }
run il_default X=1
run il_off EC3D_SAV_IL=0
run il_768 EC3D_NBLK_SPMV=768
run il_1280 EC3D_NBLK_SPMV=1280
run il_1536 EC3D_NBLK_SPMV=1536
run il_2048 EC3D_NBLK_SPMV=2048
