#!/bin/bash
# round 5, call aa: the whole GPU suite with the X groups on a second stream available (off by default), smoke
set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_aa_suite.log 2>&1
rc=$?; tail -n 12 gpurun_out/r05_aa_suite.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python __graft_entry__.py --smoke > gpurun_out/r05_aa_smoke.log 2>&1
rc=$?; tail -n 3 gpurun_out/r05_aa_smoke.log; exit $rc
