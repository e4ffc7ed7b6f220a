#!/bin/bash
# round 6, call w: the work vectors' placement -- five handles alive together without the probe (three rounds), then with it;
# then the 512^3 fixtures (bitwise) with the probe in force, and fresh processes of the bench's headline
out=$(pwd)/gpurun_out/r06w; mkdir -p $out
EC3D_PLACE_VEC=0 timeout -k 10 400 python tools/vec_place_probe.py 5 > $out/handles_no_probe.log 2>&1 || { tail $out/handles_no_probe.log; exit 1; }
cat $out/handles_no_probe.log | grep -v amdgpu.ids
EC3D_PLACE_VERBOSE=1 timeout -k 10 400 python tools/vec_place_probe.py 3 > $out/handles_probe.log 2>&1 || { tail $out/handles_probe.log; exit 1; }
cat $out/handles_probe.log | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests/test_gpu_config4.py tests/test_gpu_parity.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 4 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit 1
for i in 1 2 3 4; do
  for pv in 0 4; do
    EC3D_PLACE_VEC=$pv timeout -k 10 300 python3 tools/ab_perf.py cube512 place_vec=$pv 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
