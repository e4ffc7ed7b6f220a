#!/bin/bash
# round 4, call ae: hardware counters of the three launches of the 512^3 iteration (separate passes per group)
REPO=$(pwd); out=$REPO/gpurun_out/r04ae; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MeanOccupancyPerCU MemUnitStalled" "LdsBankConflict SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c512_$i -- python3 $REPO/bench.py --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 8 --warmup 4 > $out/c512_$i.json 2> $out/c512_$i.err || echo "pass $i ($grp) failed"
done
cd $REPO
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/r04ae"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(f"{out}/c512_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if any(s in k for s in ("k23_", "k51_", "k4s_", "k_spmv")):
            acc[k[:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())})
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
