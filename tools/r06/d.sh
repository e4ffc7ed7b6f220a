#!/bin/bash
# round 6, call d: HBM traffic of the structured SpMV kernels at 256^3 on 2-D tiles of two shapes against the linear tiles
out=$(pwd)/gpurun_out/r06d; mkdir -p $out; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
prof() { # label, counters, env...
    local label=$1 ctr=$2; shift 2
    env "$@" timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$label -- python3 $REPO/tools/av256_perf.py $label > $out/$label.log 2> $out/$label.err || { echo "$label failed"; tail -5 $out/$label.err; }
    echo "== $label $ctr"; python3 $REPO/tools/pmc_avg.py $out/$label spmv | tee -a $out/summary.log
}
export EC3D_SAV_IL=0 EC3D_FUSE23=0 EC3D_FUSE51=0
prof patch128_fetch FETCH_SIZE EC3D_SAV_PATCH=1
prof patch64_fetch FETCH_SIZE EC3D_SAV_PATCH=1 EC3D_SAV_PATCH_PX=64
prof patch32_fetch FETCH_SIZE EC3D_SAV_PATCH=1 EC3D_SAV_PATCH_PX=32
cd $REPO
for v in "EC3D_SAV_PATCH=1" "EC3D_SAV_PATCH=1 EC3D_SAV_PATCH_PX=64" "EC3D_SAV_PATCH=1 EC3D_SAV_PATCH_PX=32" "EC3D_SAV_PATCH=0"; do
  env $v timeout -k 10 240 python tools/av256_perf.py "$v" >> $out/perf.log 2>> $out/perf.err; tail -n 1 $out/perf.log
done
