#!/bin/bash
# round 5, call a: the new slab plans against the multi-rank twin, regressions of the multi / parity suites, config 4 pinned
set -o pipefail
mkdir -p gpurun_out/r05a
python -m pytest tests/test_gpu_slab_plans.py -x -q -s > gpurun_out/r05a/slab_plans.log 2>&1; echo "slab_plans rc=$?" | tee -a gpurun_out/r05a/summary.log
python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_default_policies.py -x -q > gpurun_out/r05a/regress.log 2>&1; echo "regress rc=$?" | tee -a gpurun_out/r05a/summary.log
python -m pytest tests/test_gpu_config4.py -x -q -s -k "reference_solver or slab_shapes" > gpurun_out/r05a/config4.log 2>&1; echo "config4 rc=$?" | tee -a gpurun_out/r05a/summary.log
tail -5 gpurun_out/r05a/slab_plans.log gpurun_out/r05a/regress.log gpurun_out/r05a/config4.log
