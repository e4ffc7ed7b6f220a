#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -n 40 $out/pytest.log; echo "tests failed: no timing"; exit 1; }
tail -n 2 $out/pytest.log
for wl in cube512 cube256; do
timeout -k 10 300 python tools/vec_sweep.py $wl ";NBLK_SPMV=512;NBLK_SPMV=768;NBLK_SPMV=1024;NBLK_SPMV=1280;NBLK_SPMV=1536;NBLK_SPMV=2048;PATCH=0;" > $out/pf_$wl.log 2>> $out/sweep.err
cat $out/pf_$wl.log
done
