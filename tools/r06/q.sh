#!/bin/bash
# round 6, call q: patch_pair with every request of a step out before the first wait (raw operands, forms behind the LDS
# exchange) against the previous build (tools/abtmp/libec3d_hip_base.so), same box: bitwise tests, then 512^3 and 256^3
out=$(pwd)/gpurun_out/r06q; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_plans.py tests/test_gpu_default_policies.py -q -m gpu -x > $out/pytest.log 2>&1
rc=$?; tail -n 5 $out/pytest.log | cut -c1-300; [ $rc -ne 0 ] && exit 1
for rep in 1 2; do
  for wl in cube512 cube256; do
    EC3D_LIB=$(pwd)/tools/abtmp/libec3d_hip_base.so timeout -k 10 300 python3 tools/ab_perf.py $wl base 2>> $out/ab.err | tee -a $out/ab.log || exit 1
    timeout -k 10 300 python3 tools/ab_perf.py $wl raw_forms 2>> $out/ab.err | tee -a $out/ab.log || exit 1
  done
done
