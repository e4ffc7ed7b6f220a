#!/bin/bash
# round 6, call ad: the placement search with six candidates and the 3.5 % stop: eight fresh processes
out=$(pwd)/gpurun_out/r06ad; mkdir -p $out
for i in 1 2 3 4 5 6 7 8; do
  EC3D_PLACE_VERBOSE=1 timeout -k 10 300 python3 tools/ab_perf.py cube512 search6 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
grep "vector placement" $out/ab.err | awk '{print $5, $6}' | tr '\n' ' '; echo
timeout -k 10 600 python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -k "placement" 2>&1 | tail -2
