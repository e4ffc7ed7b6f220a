#!/bin/bash
# round 6, call j: the interleaved march as adopted (register requests a step ahead) -- tests, 256^3, and where it starts to pay
out=$(pwd)/gpurun_out/r06j; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_interleaved.py -x -q -m gpu > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log; [ $rc -ne 0 ] && exit 1
run() { local name=$1; shift; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; tail -n 1 $out/perf.log | cut -c1-420; }
run default X=1
run w140 EC3D_IL_W=140
run w160 EC3D_IL_W=160
run off EC3D_SAV_IL=0
for il in 2 0; do
  DICT_ONLY=1 EC3D_SAV_IL=$il timeout -k 10 300 python tools/quick_perf_av.py 3 3 3 2>&1 | tail -n 1 | cut -c1-400 | tee -a $out/av3.log
done
for il in 2 0; do
  EC3D_SAV_IL=$il timeout -k 10 300 python tools/av256_perf.py il${il}_256x256x60 256 256 60 2>&1 | tail -n 1 | cut -c1-420 | tee -a $out/cfg3.log
done
for il in 2 0; do
  EC3D_SAV_IL=$il timeout -k 10 300 python tools/av256_perf.py il${il}_256x256x128 256 256 128 2>&1 | tail -n 1 | cut -c1-420 | tee -a $out/cfg3.log
done
