#!/bin/bash
# round 5, call ap: K2's S.S partials collapsed by K3's finalize launch (one launch fewer per iteration on five-launch slabs): suite, then ranks
set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_ap_suite.log 2>&1
rc=$?; tail -n 4 gpurun_out/r05_ap_suite.log; [ $rc -eq 0 ] || { grep -E "^FAILED|Error" gpurun_out/r05_ap_suite.log | head; exit $rc; }
for i in 1 2; do
  REHEARSE_ONLY="512,512,8,4;384,384,8,4;256,256,8,3" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_ap.log || exit 1
  REHEARSE_AV=lim timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "rank . of" >> gpurun_out/r05_ap.log || exit 1
done
cut -c1-250 gpurun_out/r05_ap.log
