#!/bin/bash
# round 4, call v: kernel trace of the deferred-X K4 variants at 512^3 (off / applying launches separately)
out=gpurun_out/r04v; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for d in 2 4; do
  EC3D_XDEFER=$d timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/d$d -o t -- python3 $GRAFT_REPO_ROOT/tools/ab_perf.py cube512 xdefer$d > $GRAFT_REPO_ROOT/$out/run_d$d.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$out/d$d -name "*kernel_stats.csv" | head -1)
  echo "== D=$d"; cut -d, -f1-5 "$f" | head -12
  find $GRAFT_REPO_ROOT/$out/d$d -name "*kernel_trace.csv" -delete
done
