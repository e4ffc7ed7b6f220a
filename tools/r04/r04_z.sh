#!/bin/bash
# round 4, call z: the deferred X update on the five-launch iteration: parity, then A/B at the mid sizes (A-V 21 M, configs
# 3 and 5, 256^3, 384^3), classic against depth 4 (and 2), same box
out=gpurun_out/r04z; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deferred" > $out/pytest.log 2>&1; rc=$?
tail -n 5 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
run() { label=$1; wl=$2; shift 2; env "$@" timeout -k 10 200 python3 tools/ab_perf.py $wl $label >> $out/ab.log 2>> $out/ab.err; }
for rep in 1 2; do
for wl in av3 lim hole cube256 cube384; do
  for d in 1 4 2; do run xdefer$d $wl EC3D_XDEFER=$d; done
done
done
cat $out/ab.log
