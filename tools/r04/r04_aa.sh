#!/bin/bash
# round 4, call aa: K4 as an SpMV kernel (k4s_x_r_spmv: A S computed again instead of written by K23 and read back):
# parity, then 512^3 and 512x512x256 with and without it, same box
out=gpurun_out/r04aa; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_default_policies.py -x -q -m gpu -k "deferred or default_policy" > $out/pytest.log 2>&1; rc=$?
tail -n 5 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in cube512 box:512x512x256; do
  run k4s0_d4 $wl EC3D_K4S=0
  run k4s1_d4 $wl EC3D_K4S=2
  run k4s1_d1 $wl EC3D_K4S=2 EC3D_XDEFER=1
  run k4s1_d2 $wl EC3D_K4S=2 EC3D_XDEFER=2
done
done
cat $out/ab.log
