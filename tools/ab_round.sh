#!/bin/bash
tag=${1:-ab}; out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timeloop.py tests/test_gpu_shipped_inputs.py -m gpu -q -x > $out/pytest.log 2>&1 || { tail -n 60 $out/pytest.log; echo "tests failed: no timing"; exit 1; }
tail -n 2 $out/pytest.log
t() { timeout -k 10 240 "$@" >> $out/ab.log 2>> $out/ab.err || { echo "rc=$? $*" >> $out/ab.log; }; }
for rep in 1 2 3; do t python tools/ab_perf.py av3 two_slots_yz; done
cat $out/ab.log
