#!/bin/bash
# round 5, call w: where is the GPU idle inside a rank's iteration?  kernel timeline of rank 4 of 8 (512^3) and rank 2 of 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "512,512,8,4" "512,512,4,2"; do
  tag=$(echo $cfg | tr , _)
  rm -rf $R/gpurun_out/prof_w_$tag
  REHEARSE_ONLY="$cfg" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_w_$tag -- python3 $R/tools/rank_rehearsal.py 60 > $R/gpurun_out/r05_w_$tag.log 2>&1 || { tail -n 20 $R/gpurun_out/r05_w_$tag.log; exit 1; }
  f=$(find $R/gpurun_out/prof_w_$tag -name '*kernel_trace.csv' | head -n 1)
  echo "== $cfg" >> $R/gpurun_out/r05_w.log
  grep "ms per iteration" $R/gpurun_out/r05_w_$tag.log >> $R/gpurun_out/r05_w.log
  python3 $R/tools/timeline_gaps.py $f k4 >> $R/gpurun_out/r05_w.log 2>&1
  rm -rf $R/gpurun_out/prof_w_$tag
done
cat $R/gpurun_out/r05_w.log
