#!/bin/bash
out=gpurun_out/r04l; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_formats_dist.py tests/test_gpu_default_policies.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $out/pytest.log
for rep in 1 2; do for wl in lim hole av3; do timeout -k 10 100 python3 tools/ab_perf.py $wl r04_rows_eff >> $out/ab.log 2>> $out/ab.err; done; done
cat $out/ab.log
timeout -k 10 400 python3 tools/config_runs.py --mode overlap > $out/config_runs.log 2> $out/config_runs.err; cat $out/config_runs.log
