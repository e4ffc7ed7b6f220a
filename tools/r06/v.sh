#!/bin/bash
# round 6, call v: a K1 stage on a handle nobody has set up (state initialised at creation now) -- the test alone, then in its file
out=$(pwd)/gpurun_out/r06v; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_formats_dist.py -q -m gpu -k "slab_spmv_equals" 2>&1 | tail -3
