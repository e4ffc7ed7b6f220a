#!/bin/bash
# One gpurun call: the GPU test suite, then the bench lines of the three workloads.  A step that times out ends the call.
#   tools/gpu_round.sh <tag> [pytest args...]
tag=${1:-run}; shift
out=gpurun_out/$tag
mkdir -p $out
run() { # name, seconds, command...
    local name=$1 secs=$2; shift 2
    echo "== $name" | tee -a $out/progress.log
    timeout -k 10 $secs "$@" > $out/$name.log 2>&1
    local rc=$?
    echo "== $name rc=$rc" | tee -a $out/progress.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping" | tee -a $out/progress.log; exit $rc; fi
    return $rc
}
run pytest 900 python -m pytest tests -m gpu -q -x -rsPx "$@"
tail -n 5 $out/pytest.log
run bench_cube 300 python bench.py --no-cpu-baseline
tail -n 1 $out/bench_cube.log | cut -c1-600
run bench_av 300 python bench.py --workload av --no-cpu-baseline
tail -n 1 $out/bench_av.log | cut -c1-600
run bench_dia 300 python bench.py --format dia --no-cpu-baseline
tail -n 1 $out/bench_dia.log | cut -c1-600
