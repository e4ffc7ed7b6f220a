#!/bin/bash
# round 6, call an (second run): interleaved A-V march -- the one-sided face slots requested a step ahead (class bytes two steps ahead)
out=$(pwd)/gpurun_out/r06an; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_interleaved.py tests/test_gpu_av256.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 3 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && { grep -n "^E " $out/pytest.log | head -20; exit 1; }
for i in 1 2 3; do
  EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_il0.so timeout -k 10 400 python3 tools/av256_perf.py before 2>> $out/av.err | tail -n 1 | cut -c1-20,150-420 | tee -a $out/av.log
  timeout -k 10 400 python3 tools/av256_perf.py face_slots_ahead 2>> $out/av.err | tail -n 1 | cut -c1-20,150-420 | tee -a $out/av.log
done
