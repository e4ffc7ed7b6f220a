#!/bin/bash
# round 4, call x: deferred-X K4: eight tiles in flight in the launch without X; depths 3 and 4; kernel trace of the best
out=gpurun_out/r04x; mkdir -p $out
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
run classic cube512 EC3D_XDEFER=1
for d in 2 3 4; do
  for off in 4 8; do for on in 1 2; do
    run d${d}_off${off}_on${on} cube512 EC3D_XDEFER=$d EC3D_XD_OFF_DEPTH=$off EC3D_XD_ON_DEPTH=$on
  done; done
done
run classic cube512 EC3D_XDEFER=1
cat $out/ab.log
cd /tmp && export TMPDIR=/tmp
EC3D_XDEFER=4 EC3D_XD_OFF_DEPTH=8 EC3D_XD_ON_DEPTH=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -o t -- python3 $GRAFT_REPO_ROOT/tools/ab_perf.py cube512 traced > $GRAFT_REPO_ROOT/$out/trace_run.log 2>&1
f=$(find $GRAFT_REPO_ROOT/$out/trace -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 "$f" | head -8
find $GRAFT_REPO_ROOT/$out/trace -name "*kernel_trace.csv" -delete
