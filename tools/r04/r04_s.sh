#!/bin/bash
# round 4, call s: per-rank overlapped output of the one-process-per-GPU host (host._SlabOutputPipeline), the slot guard
# of ec3d_vtk_fields_wait, the stoppable sources thread
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_two_process.py tests/test_vtk_output.py tests/test_gpu_host_program.py tests/test_gpu_multi.py \
    tests/test_host_sources.py -x -q -m gpu > gpurun_out/r04s_tests.log 2>&1 || { tail -40 gpurun_out/r04s_tests.log; exit 1; }
tail -3 gpurun_out/r04s_tests.log
