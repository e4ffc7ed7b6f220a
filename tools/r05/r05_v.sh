#!/bin/bash
# round 5, call v: rank handles over the loopback transport -- time loop state, field output, CSR route
timeout -k 10 400 python -m pytest tests/test_gpu_rank_loopback.py -x -q > gpurun_out/r05_v.log 2>&1; rc=$?
tail -n 40 gpurun_out/r05_v.log; exit $rc
