#!/bin/bash
# round 6, call aj: three launches from 20 Mi rows on an undivided handle: the default-policy tests (bitwise), parity, contract
out=$(pwd)/gpurun_out/r06aj; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_default_policies.py tests/test_gpu_parity.py tests/test_bench_contract.py tests/test_gpu_edge_cases.py tests/test_gpu_config4.py tests/test_gpu_multi.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && { grep -n "^E " $out/pytest.log | head -20; exit 1; }
for g in 512x512x80 512x512x96 512x512x112; do timeout -k 10 300 python3 tools/ab_perf.py box:$g default_now 2>> $out/ab.err | cut -c1-150; done
