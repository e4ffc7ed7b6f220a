#!/bin/bash
# round 6, call c: HBM and L2 traffic of the structured SpMV kernels at 256^3, separate U list against the interleaved march
out=$(pwd)/gpurun_out/r06c; mkdir -p $out; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
prof() { # label, counters, env...
    local label=$1 ctr=$2; shift 2
    env "$@" timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$label -- python3 $REPO/tools/av256_perf.py $label > $out/$label.log 2> $out/$label.err || { echo "$label failed"; tail -5 $out/$label.err; }
    echo "== $label $ctr"; python3 $REPO/tools/pmc_avg.py $out/$label spmv | tee -a $out/summary.log
}
prof off_fetch FETCH_SIZE EC3D_SAV_IL=0
prof il_fetch FETCH_SIZE EC3D_SAV_IL=1
prof il2048_fetch FETCH_SIZE EC3D_SAV_IL=1 EC3D_NBLK_SPMV=2048
prof off_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" EC3D_SAV_IL=0
prof il_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" EC3D_SAV_IL=1
prof off_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" EC3D_SAV_IL=0
prof il_tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" EC3D_SAV_IL=1
