#!/bin/bash
# round 6, call l: every form of bench.py under test
out=$(pwd)/gpurun_out/r06l; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_bench_forms.py tests/test_bench_contract.py -q -m gpu -x --durations=12 > $out/pytest.log 2>&1
rc=$?; tail -n 40 $out/pytest.log | cut -c1-400; exit $rc
