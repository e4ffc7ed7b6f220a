#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q -s -k "resampled or both_splits" > gpurun_out/r05_at.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r05_at.log | tail -n 12; exit $rc
