#!/bin/bash
out=gpurun_out/r04f; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rsx > $out/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 8 $out/pytest.log
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 100 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for wl in av1 lim0 hole0; do
  run r03 $wl EC3D_LIB=tools/ab/libec3d_hip_r03.so
  run r04 $wl A=1
done
cat $out/ab.log
