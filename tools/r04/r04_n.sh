#!/bin/bash
out=gpurun_out/r04n; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rsx > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 4 $out/pytest.log
timeout -k 10 400 python3 tools/config_runs.py --mode overlap > $out/config_runs.log 2> $out/config_runs.err; cat $out/config_runs.log
for rep in 1 2; do timeout -k 10 100 python3 tools/ab_perf.py lim final >> $out/ab.log 2>> $out/ab.err; done; cat $out/ab.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json
