#!/bin/bash
# a few hardware counters of the SpMV kernels: the structured A-V system against the 256^3 cube (separate passes per group)
REPO=$(pwd); out=$REPO/gpurun_out/r04p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "MeanOccupancyPerCU MemUnitStalled" "LdsBankConflict SQ_INSTS_LDS SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/av_$i -- python3 $REPO/bench.py --workload av --no-cpu-baseline --no-side-workloads --steps 3 --warmup 1 > $out/av_$i.json 2> $out/av_$i.err || echo "av pass $i failed"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c256_$i -- python3 $REPO/bench.py --grid 256 --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 3 --warmup 1 > $out/c256_$i.json 2> $out/c256_$i.err || echo "256 pass $i failed"
done
cd $REPO
python3 - <<'PY'
import csv, glob, os, collections
out = "gpurun_out/r04p"
for wl in ("av", "c256"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{wl}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(s in k for s in ("k1_spmv", "k3_spmv", "k4_x", "k2_s", "k5_p")):
                acc[k[:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(wl, k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())})
PY
