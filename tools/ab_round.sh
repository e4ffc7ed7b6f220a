#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
t() { timeout -k 10 240 "$@" >> $out/ab.log 2>> $out/ab.err || { echo "rc=$? $*" >> $out/ab.log; }; }
EC3D_BAND_TILED=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats_dist.py tests/test_gpu_edge_cases.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -n 30 $out/pytest.log; echo "tests failed: no timing"; exit 1; }
tail -n 2 $out/pytest.log
for rep in 1 2 3 4 5 6; do
  EC3D_BAND_TILED=1 t python tools/ab_perf.py dia512 tiled
  t python tools/ab_perf.py dia512 streams
done
EC3D_BAND_TILED=1 EC3D_NBLK_SPMV=1024 t python tools/ab_perf.py dia512 tiled_1024
EC3D_BAND_TILED=1 EC3D_NBLK_SPMV=512 t python tools/ab_perf.py dia512 tiled_512
EC3D_BAND_TILED=1 EC3D_DIA_NT=0 t python tools/ab_perf.py dia512 tiled_plainloads
cat $out/ab.log
