#!/bin/bash
out=gpurun_out/r04i; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rsx > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 9 $out/pytest.log
timeout -k 10 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json
python __graft_entry__.py --smoke 2>&1 | tail -2
