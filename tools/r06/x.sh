#!/bin/bash
# round 6, call x: the placement search under test (edge cases file), then BASELINE config 3 at 256^3 with and without it
out=$(pwd)/gpurun_out/r06x; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_edge_cases.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && { grep -n "^E " $out/pytest.log | head -20; exit 1; }
for i in 1 2 3; do
  for pv in 0 4; do
    EC3D_PLACE_VEC=$pv EC3D_PLACE_VERBOSE=1 timeout -k 10 400 python3 tools/av256_perf.py 2>> $out/av.err | tail -n 3 | sed "s/^/place_vec=$pv /" | tee -a $out/av.log
  done
done
grep "vector placement" $out/av.err | tail -12
