#!/bin/bash
# round 5, call k: where do the A-V SpMV kernels' extra fetches come from -- HBM or the Infinity Cache?  No counter of this
# box separates them (TCC_EA0_RDREQ_DRAM == TCC_EA0_RDREQ: the Infinity Cache sits behind the L2's memory port), but the mean
# latency of an L2 read miss does: TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ (requests in flight per cycle / requests per cycle).
# Yardsticks in the same call: the 512^3 cube (8.6 GB of vectors: every miss goes to HBM) and a 128^3 cube (everything
# lives in the Infinity Cache).
REPO=$(pwd); out=$REPO/gpurun_out/r05k; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/av_$i -- python3 $REPO/bench.py --workload av --no-cpu-baseline --no-side-workloads --steps 3 --warmup 1 > $out/av_$i.json 2> $out/av_$i.err || echo "av pass $i failed"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c256_$i -- python3 $REPO/bench.py --grid 256 --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 3 --warmup 1 > $out/c256_$i.json 2> $out/c256_$i.err || echo "256 pass $i failed"
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c128_$i -- python3 $REPO/bench.py --grid 128 --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 3 --warmup 1 > $out/c128_$i.json 2> $out/c128_$i.err || echo "128 pass $i failed"
  EC3D_FUSE23=0 EC3D_FUSE51=0 timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/c512_$i -- python3 $REPO/bench.py --grid 512 --no-cpu-baseline --no-side-workloads --no-spmv-dia --steps 3 --warmup 1 > $out/c512_$i.json 2> $out/c512_$i.err || echo "512 pass $i failed"
done
cd $REPO
python3 - <<'PY' | tee gpurun_out/r05k/table.log
import csv, glob, os, collections
out = "gpurun_out/r05k"
for wl in ("av", "c256", "c128", "c512"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for fn in glob.glob(f"{out}/{wl}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if any(s in k for s in ("k1_spmv", "k3_spmv", "k4_x", "k4d_x", "k2_s", "k5_p")):
                acc[k[:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        m = {c: sum(v) / len(v) for c, v in acc[k].items()}
        lat = m.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / max(m.get("TCC_EA0_RDREQ_sum", 1), 1)
        print(wl, k, f"mean L2-miss read latency {lat:.0f} L2 cycles;", {c: round(v, 1) for c, v in sorted(m.items())})
PY
rm -rf gpurun_out/r05k/*_[0-9]   # the raw traces stay on the box
