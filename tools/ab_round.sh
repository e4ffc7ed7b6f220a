#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -n 60 $out/pytest.log; echo "tests failed: no timing"; exit 1; }
tail -n 2 $out/pytest.log
timeout -k 10 300 python tools/vec_sweep.py cube512 ";FUSE51=0,FUSE23=0;NBLK_SPMV=512;NBLK_SPMV=1536;FUSE51=0,FUSE23=0,NBLK_SPMV=1536;" > $out/ahead_512.log 2>> $out/err.log
cat $out/ahead_512.log
timeout -k 10 300 python tools/vec_sweep.py cube256 ";NBLK_SPMV=768;NBLK_SPMV=1024;NBLK_SPMV=1280" > $out/ahead_256.log 2>> $out/err.log
cat $out/ahead_256.log
