#!/bin/bash
# round 5, call r: the rank driver with several ranks on one GPU through the loopback transport, then the whole GPU suite and the smoke
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_rank_loopback.py -x -q -s > gpurun_out/r05_r_loopback.log 2>&1
rc=$?; echo "loopback rc=$rc" | tee -a gpurun_out/r05_r_loopback.log
tail -n 8 gpurun_out/r05_r_loopback.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05_r_suite.log 2>&1
rc=$?; echo "suite rc=$rc" | tee -a gpurun_out/r05_r_suite.log
tail -n 8 gpurun_out/r05_r_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r05_r_smoke.log 2>&1
rc=$?; echo "smoke rc=$rc" | tee -a gpurun_out/r05_r_smoke.log
tail -n 5 gpurun_out/r05_r_smoke.log
exit $rc
