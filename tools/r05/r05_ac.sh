#!/bin/bash
# round 5, call ac: kernel timeline of rank 4 of 8 (512^3) on plans 2 and 1: what do the split K2 / K5 launches cost?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for plan in 2 1; do
  rm -rf $R/gpurun_out/prof_ac
  EC3D_SLAB_PLAN=$plan REHEARSE_ONLY="512,512,8,4" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ac -- python3 $R/tools/rank_rehearsal.py 60 > $R/gpurun_out/r05_ac_$plan.log 2>&1 || { tail -n 20 $R/gpurun_out/r05_ac_$plan.log; exit 1; }
  f=$(find $R/gpurun_out/prof_ac -name '*kernel_trace.csv' | head -n 1)
  echo "== plan $plan" >> $R/gpurun_out/r05_ac.log
  grep "ms per iteration" $R/gpurun_out/r05_ac_$plan.log >> $R/gpurun_out/r05_ac.log
  python3 $R/tools/timeline_gaps.py $f k4d >> $R/gpurun_out/r05_ac.log 2>&1
  python3 - $f >> $R/gpurun_out/r05_ac.log <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1])))
rows = rows[len(rows) * 3 // 4:]
# one iteration in the steady part: from one k5 to the next
idx = [i for i, r in enumerate(rows) if "k1_spmv_dot" in r[2]]
a = idx[4]
t0 = rows[a][0]
for s, e, k, st in rows[a:a + 40]:
    print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  stream {st}  {k}")
PY
  rm -rf $R/gpurun_out/prof_ac
done
cat $R/gpurun_out/r05_ac.log
