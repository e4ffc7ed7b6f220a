#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel trace + the two PMC passes for bench.py.
# Usage: tools/profile_bench.sh <tag> [grid] [dict|dia] [cube|av] [refine]
set -e
TAG=${1:-r01}; GRID=${2:-512}; FMT=${3:-dict}; WL=${4:-cube}; REF=${5:-3}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--grid $GRID --no-cpu-baseline --no-side-workloads --format $FMT --workload $WL --refine $REF"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $ARGS --steps 30 --warmup 5 > $OUT/bench_trace.json 2> $OUT/trace.err || { tail -20 $OUT/trace.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py $ARGS --no-spmv-dia --steps 8 --warmup 4 > $OUT/bench_fetch.json 2> $OUT/fetch.err || { tail -20 $OUT/fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py $ARGS --no-spmv-dia --steps 8 --warmup 4 > $OUT/bench_write.json 2> $OUT/write.err || { tail -20 $OUT/write.err; exit 1; }
cd $REPO
python3 tools/parse_rocprof.py $OUT $TAG $GRID $FMT $WL
