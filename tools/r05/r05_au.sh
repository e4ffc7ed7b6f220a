#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_rank_loopback.py -x -q -k "uneven" > gpurun_out/r05_au.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r05_au.log | tail -n 15; exit $rc
