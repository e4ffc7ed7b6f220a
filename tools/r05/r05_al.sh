#!/bin/bash
# round 5, call al: conductor-free U planes stay at home: A-V slab tests, then configs 5 / 3 ranks with and without
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_rank_loopback.py tests/test_gpu_timeloop.py tests/test_gpu_rccl_rank.py tests/test_gpu_formats_dist.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r05_al_tests.log 2>&1
rc=$?; tail -n 5 gpurun_out/r05_al_tests.log; [ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
  echo "== EC3D_AV_SEND_EMPTY_U=$v" >> gpurun_out/r05_al.log
  EC3D_AV_SEND_EMPTY_U=$v REHEARSE_AV=all timeout -k 10 400 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_al.log || exit 1
done
cut -c1-260 gpurun_out/r05_al.log
