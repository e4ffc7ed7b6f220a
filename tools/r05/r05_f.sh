#!/bin/bash
# round 5, call f: what the four partial-sum collapse launches cost a rank (timing experiment: sums wrong on purpose)
set -o pipefail
mkdir -p gpurun_out/r05f
for skip in 0 1 0 1; do
  echo "== skip finalize: $skip" >> gpurun_out/r05f/finalize.log
  if [ $skip = 1 ]; then export EC3D_TIMING_SKIP_FINALIZE=1; else unset EC3D_TIMING_SKIP_FINALIZE; fi
  REHEARSE_ONLY="512,512,8,4;256,256,8,3;64,128,8,3" timeout -k 10 200 python tools/rank_rehearsal.py 300 >> gpurun_out/r05f/finalize.log 2>&1
done
grep -v "version\|Hostname\|Librccl\|amdgpu.ids" gpurun_out/r05f/finalize.log
