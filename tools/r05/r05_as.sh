#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_rccl_rank.py tests/test_gpu_multi.py tests/test_gpu_rank_loopback.py tests/test_gpu_config4.py tests/test_gpu_dropin_reference_program.py -x -q > gpurun_out/r05_as.log 2>&1
rc=$?; tail -n 5 gpurun_out/r05_as.log; [ $rc -eq 0 ] || exit $rc
REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "through the RCCL driver" | cut -c1-260
