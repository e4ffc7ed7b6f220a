#!/bin/bash
# round 6, call e: where the structured SpMV's extra fetches come from -- cube on linear tiles / 2-D tiles, A-V without a conductor
out=$(pwd)/gpurun_out/r06e; mkdir -p $out; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
prof() { # label, script, counters, env...
    local label=$1 script=$2 ctr=$3; shift 3
    env "$@" timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$label -- python3 $REPO/tools/$script $label > $out/$label.log 2> $out/$label.err || { echo "$label failed"; tail -5 $out/$label.err; }
    echo "== $label $ctr"; python3 $REPO/tools/pmc_avg.py $out/$label spmv | tee -a $out/summary.log; tail -n 1 $out/$label.log | cut -c1-300
}
export EC3D_SAV_IL=0 EC3D_FUSE23=0 EC3D_FUSE51=0
prof cube_linear cube_perf.py FETCH_SIZE EC3D_PATCH=0
prof cube_patch cube_perf.py FETCH_SIZE EC3D_PATCH=1
prof av_air av256_perf.py FETCH_SIZE AIR=1
prof av_nt0 av256_perf.py FETCH_SIZE EC3D_NT=0
prof av_il_air av256_perf.py FETCH_SIZE AIR=1 EC3D_SAV_IL=1
