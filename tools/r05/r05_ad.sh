#!/bin/bash
# round 5, call ad: plan 5 (K2 / K5 boundary first AND K1 / K3 interior first): parity, then rank 4 of 8 against plans 1 and 2
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_slab_plans.py tests/test_gpu_rank_loopback.py -x -q > gpurun_out/r05_ad_tests.log 2>&1
rc=$?; tail -n 6 gpurun_out/r05_ad_tests.log; [ $rc -eq 0 ] || exit $rc
for v in 1 5 2 1 5 2; do
  echo "== EC3D_SLAB_PLAN=$v" >> gpurun_out/r05_ad.log
  EC3D_SLAB_PLAN=$v REHEARSE_ONLY="512,512,8,4;384,384,8,4;256,256,8,3" timeout -k 10 300 python3 tools/rank_rehearsal.py 200 2>&1 | grep "ms per iteration" >> gpurun_out/r05_ad.log || exit 1
done
cut -c1-300 gpurun_out/r05_ad.log
