#!/bin/bash
# round 5, call l: are the structured A-V SpMV kernels short of memory or of instruction issue?  VALU / SALU / LDS activity
# of k1 / k3 <207> at 21.4 M unknowns beside the cube kernels <107> at 256^3 (separate passes per counter group)
REPO=$(pwd); out=$REPO/gpurun_out/r05l; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "VALUBusy SALUBusy VALUUtilization" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/av_$i -- python3 $REPO/bench.py --workload av --no-cpu-baseline --no-side-workloads --steps 3 --warmup 1 > $out/av_$i.json 2> $out/av_$i.err || echo "av pass $i failed"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c256_$i -- python3 $REPO/bench.py --grid 256 --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 3 --warmup 1 > $out/c256_$i.json 2> $out/c256_$i.err || echo "256 pass $i failed"
done
cd $REPO
python3 - <<'PY' | tee gpurun_out/r05l/table.log
import csv, glob, os, collections
out = "gpurun_out/r05l"
for wl in ("av", "c256"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{wl}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(s in k for s in ("k1_spmv", "k3_spmv", "k2_s", "k5_p")):
                acc[k[:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        print(wl, k, {c: round(v, 1) for c, v in sorted(m.items())})
PY
rm -rf gpurun_out/r05l/*_[0-9]
