#!/bin/bash
# round 6, call m: instruction activity per tile kind (structured SpMV kernels at 256^3, with / without the conductor, list / interleaved),
# and of the three launches of the 512^3 iteration (K23 beside K51 and K4s)
out=$(pwd)/gpurun_out/r06m; mkdir -p $out; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
CTR="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
CTR2="SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
prof() { # label, counters, command...
    local label=$1 ctr=$2; shift 2
    timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$label -- "$@" > $out/$label.log 2> $out/$label.err || { echo "$label failed"; tail -5 $out/$label.err; }
    echo "== $label"; python3 $REPO/tools/pmc_avg.py $out/$label spmv | tee -a $out/summary.log
}
AIR=1 EC3D_SAV_IL=0 prof air_list "$CTR" python3 $REPO/tools/av256_perf.py air_list
AIR=1 EC3D_SAV_IL=1 prof air_il "$CTR" python3 $REPO/tools/av256_perf.py air_il
EC3D_SAV_IL=0 prof cond_list "$CTR" python3 $REPO/tools/av256_perf.py cond_list
EC3D_SAV_IL=1 prof cond_il "$CTR" python3 $REPO/tools/av256_perf.py cond_il
EC3D_SAV_IL=0 prof cond_list2 "$CTR2" python3 $REPO/tools/av256_perf.py cond_list2
EC3D_SAV_IL=1 prof cond_il2 "$CTR2" python3 $REPO/tools/av256_perf.py cond_il2
prof cube512 "$CTR" python3 $REPO/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-workloads --no-spmv-dia
prof cube512b "$CTR2" python3 $REPO/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-side-workloads --no-spmv-dia
echo "== cube512 kernels"; python3 $REPO/tools/pmc_avg.py $out/cube512 k | tee -a $out/summary.log
python3 $REPO/tools/pmc_avg.py $out/cube512b k | tee -a $out/summary.log
