#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rank_loopback.py tests/test_gpu_rccl_rank.py tests/test_gpu_multi.py tests/test_gpu_config4.py tests/test_gpu_timeloop.py -x -q > gpurun_out/r05_af.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_af.log; exit $rc
