#!/bin/bash
# round 6, call ad (third run): six candidates, the losing sets held until the search is over: eight fresh processes, what it costs; then larger grids
out=$(pwd)/gpurun_out/r06ad; mkdir -p $out
for i in 1 2 3 4 5 6 7 8; do
  EC3D_PLACE_VERBOSE=1 timeout -k 10 300 python3 tools/ab_perf.py cube512 search6_hold 2>> $out/ab.err | tee -a $out/ab.log || exit 1
done
grep "vector placement" $out/ab.err | awk '{print $5, $6}' | tr '\n' ' '; echo
timeout -k 10 600 python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -k "placement" 2>&1 | tail -2
bash tools/r06/ah.sh 2>&1 | grep -E "allocated in [0-9]{2,}|candidates"
