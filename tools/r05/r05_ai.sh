#!/bin/bash
timeout -k 10 300 python3 tools/av_plan5_check.py 64 32 48 2 > gpurun_out/r05_ai_check.log 2>&1; rc=$?
timeout -k 10 300 python3 tools/av_plan5_check.py 96 40 36 3 >> gpurun_out/r05_ai_check.log 2>&1; rc2=$?
grep -v amdgpu.ids gpurun_out/r05_ai_check.log | cut -c1-600 | tail -n 14
[ $rc -eq 0 ] && [ $rc2 -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_rank_loopback.py tests/test_gpu_timeloop.py tests/test_gpu_rccl_rank.py -q > gpurun_out/r05_ai_tests.log 2>&1
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r05_ai_tests.log | tail -n 20
