#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py tests/test_gpu_edge_cases.py tests/test_bench_contract.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -n 40 $out/pytest.log; echo "tests failed: no timing"; exit 1; }
tail -n 2 $out/pytest.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $out/bench.log 2>&1; tail -n 1 $out/bench.log | cut -c1-1500
