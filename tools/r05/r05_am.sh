#!/bin/bash
# round 5, call am: kernel timeline of rank 2 of 4 of the 512^3 job (plan 4, three launches)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_am
REHEARSE_ONLY="512,512,4,2" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_am -- python3 $R/tools/rank_rehearsal.py 60 > $R/gpurun_out/r05_am_run.log 2>&1 || { tail -n 20 $R/gpurun_out/r05_am_run.log; exit 1; }
f=$(find $R/gpurun_out/prof_am -name '*kernel_trace.csv' | head -n 1)
grep "ms per iteration" $R/gpurun_out/r05_am_run.log > $R/gpurun_out/r05_am.log
python3 - $f >> $R/gpurun_out/r05_am.log <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if "k23_s_spmv_dots" in r[2]]
a = idx[len(idx) * 2 // 3]
t0 = rows[a][0]
for s, e, k, st in rows[a:a + 60]:
    print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  stream {st}  {k}")
PY
rm -rf $R/gpurun_out/prof_am
cut -c1-200 $R/gpurun_out/r05_am.log
