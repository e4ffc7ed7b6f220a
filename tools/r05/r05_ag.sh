#!/bin/bash
# round 5, call ag: kernel timeline of rank 3 of 8 of BASELINE config 5 (A-V, LIM at 384x192x128)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ag
REHEARSE_AV=lim timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_ag -- python3 $R/tools/rank_rehearsal.py 60 > $R/gpurun_out/r05_ag_run.log 2>&1 || { tail -n 20 $R/gpurun_out/r05_ag_run.log; exit 1; }
f=$(find $R/gpurun_out/prof_ag -name '*kernel_trace.csv' | head -n 1)
grep "rank . of" $R/gpurun_out/r05_ag_run.log > $R/gpurun_out/r05_ag.log
python3 - $f >> $R/gpurun_out/r05_ag.log <<'PY'
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "?")) for r in csv.DictReader(open(sys.argv[1])))
# the run holds: undivided, rank 3 of 8, undivided, rank 1 of 4.  Take the second quarter's tail = rank 3 of 8's steady part
n = len(rows)
def dump(rows, label):
    idx = [i for i, r in enumerate(rows) if "k1_spmv_dot" in r[2]]
    a = idx[len(idx) // 2]
    t0 = rows[a][0]
    print("==", label)
    for s, e, k, st in rows[a:a + 46]:
        print(f"   +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  stream {st}  {k}")
# split by big time gaps (set-up between the four runs)
cuts = [0] + [i for i in range(1, n) if rows[i][0] - rows[i - 1][1] > 50_000_000] + [n]
segs = [rows[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
segs = [s for s in segs if sum(1 for r in s if "k1_spmv_dot" in r[2]) > 40]
for i, s in enumerate(segs):
    dump(s, f"segment {i} ({len(s)} kernels)")
PY
rm -rf $R/gpurun_out/prof_ag
cut -c1-200 $R/gpurun_out/r05_ag.log
