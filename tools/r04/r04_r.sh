#!/bin/bash
# round 4, call r: the overlapped field output over the slabs of the multi handle (ec3d_multi_vtk_fields_begin / _wait):
# parity of the files, then config 5 on 4 slabs of one card with every file written, overlapped against synchronous
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_host_program.py tests/test_gpu_fullsize.py -x -q -m gpu \
    -k "fields or overlapped or slabs or multi" > gpurun_out/r04r_tests.log 2>&1
tail -3 gpurun_out/r04r_tests.log
timeout -k 10 400 python tools/config_runs.py --slabs 4 --cases LIM --mode overlap > gpurun_out/r04r_lim_4slabs_overlap.log 2>&1
tail -2 gpurun_out/r04r_lim_4slabs_overlap.log
timeout -k 10 400 python tools/config_runs.py --slabs 4 --cases LIM --mode sync > gpurun_out/r04r_lim_4slabs_sync.log 2>&1
tail -2 gpurun_out/r04r_lim_4slabs_sync.log
