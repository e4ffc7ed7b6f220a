#!/bin/bash
# round 4, call u: the deferred X update (k4d_x_r_update): parity, then K4 and the iteration at 512^3 and 512x512x256 with
# the classic K4 (EC3D_XDEFER=1) against depths 2, 3, 4, same box
out=gpurun_out/r04u; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred or two_dimensional" > $out/pytest.log 2>&1; rc=$?
tail -n 15 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in cube512 box:512x512x256; do
  for d in 1 2 3 4; do run xdefer$d $wl EC3D_XDEFER=$d; done
done
done
cat $out/ab.log
