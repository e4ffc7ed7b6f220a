#!/bin/bash
# round 6, call o: the interleaved march on BASELINE config 5 (LIM at 384 x 192 x 128, 29.8 M streamed rows: just above its threshold)
out=$(pwd)/gpurun_out/r06o; mkdir -p $out
for il in 1 0; do
  STEM=LIM EC3D_SAV_IL=$il timeout -k 10 300 python tools/av256_perf.py cfg5_il$il 384 192 128 2>&1 | tail -n 1 | cut -c1-420 | tee -a $out/perf.log
done
for il in 1 0; do
  STEM=LIM EC3D_SAV_IL=$il timeout -k 10 300 python tools/av256_perf.py lim_il$il 384 192 256 2>&1 | tail -n 1 | cut -c1-420 | tee -a $out/perf.log
done
for il in 1 0; do
  EC3D_SAV_IL=$il timeout -k 10 300 python tools/av256_perf.py hole_384_il$il 384 384 128 2>&1 | tail -n 1 | cut -c1-420 | tee -a $out/perf.log
done
