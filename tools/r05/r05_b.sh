#!/bin/bash
# round 5, call b: slab plans again (restart rule, iterate numbering), the multi suites, per-rank times at the slab shapes
set -o pipefail
mkdir -p gpurun_out/r05b
python -m pytest tests/test_gpu_slab_plans.py -x -q -s > gpurun_out/r05b/slab_plans.log 2>&1; echo "slab_plans rc=$?" | tee -a gpurun_out/r05b/summary.log
python -m pytest tests/test_gpu_multi.py tests/test_gpu_formats_dist.py tests/test_gpu_timeloop.py tests/test_gpu_edge_cases.py -x -q > gpurun_out/r05b/regress.log 2>&1; echo "regress rc=$?" | tee -a gpurun_out/r05b/summary.log
python tools/slab_shapes.py 100 > gpurun_out/r05b/slab_shapes.log 2>&1; echo "slab_shapes rc=$?" | tee -a gpurun_out/r05b/summary.log
python bench.py --gpus 2 --devices 0,0 --steps 50 --no-cpu-baseline > gpurun_out/r05b/bench_2slabs.json 2> gpurun_out/r05b/bench_2slabs.err; echo "bench2 rc=$?" | tee -a gpurun_out/r05b/summary.log
tail -n 6 gpurun_out/r05b/slab_plans.log; tail -n 4 gpurun_out/r05b/regress.log; cat gpurun_out/r05b/slab_shapes.log
