#!/bin/bash
# round 4, first GPU call: the whole GPU suite (new: cache policies under the twin, default-policy sizes, restart
# counted, tree gather, no wait mutex), the bench line with the side workloads, and baseline per-kernel timings of
# the workloads this round works on (same box: r03 build against the current one).
out=gpurun_out/r04a; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rsx > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $out/pytest.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-400 $out/bench.json
for wl in av1 av3 hole lim cube256; do
  EC3D_LIB=tools/ab/libec3d_hip_r03.so timeout -k 10 200 python3 tools/ab_perf.py $wl r03 >> $out/ab.log 2>> $out/ab.err
  timeout -k 10 200 python3 tools/ab_perf.py $wl r04 >> $out/ab.log 2>> $out/ab.err
done
cat $out/ab.log
timeout -k 10 200 python3 tools/multi_host_overhead.py > $out/multi_host_overhead.log 2>&1; tail -n 12 $out/multi_host_overhead.log
