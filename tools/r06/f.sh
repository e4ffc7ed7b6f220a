#!/bin/bash
# round 6, call f: the fused interleaved step -- tests, then BASELINE config 3 at 256^3 and its fetches
out=$(pwd)/gpurun_out/r06f; mkdir -p $out; REPO=$(pwd)
timeout -k 10 600 python -m pytest tests/test_gpu_interleaved.py -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; tail -n 6 $out/pytest.log; [ $rc -ne 0 ] && exit 1
run() { local name=$1; shift; echo "== $name"; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; local rc=$?; echo "rc=$rc"; tail -n 1 $out/perf.log; [ $rc -eq 124 ] && exit 1; return 0; }
run il_default X=1
run il_off EC3D_SAV_IL=0
run il_256 EC3D_NBLK_SPMV=256
run il_384 EC3D_NBLK_SPMV=384
run il_768 EC3D_NBLK_SPMV=768
run il_1024 EC3D_NBLK_SPMV=1024
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/il_fetch -- python3 $REPO/tools/av256_perf.py il_fetch > $out/il_fetch.log 2> $out/il_fetch.err
python3 $REPO/tools/pmc_avg.py $out/il_fetch spmv
