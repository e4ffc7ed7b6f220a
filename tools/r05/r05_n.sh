#!/bin/bash
# round 5, call n: structured A-V kernels with their scalar bookkeeping in 32 bits, against the build before (EC3D_LIB), same box
set -o pipefail
out=gpurun_out/r05n; mkdir -p $out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_default_policies.py tests/test_gpu_formats_dist.py -x -q -k "struct or av or sav or runtime_shaped or csr_route or default_policy" > $out/parity.log 2>&1; echo "parity rc=$?" | tee -a $out/summary.log
for rep in 1 2 3; do
  for wl in av3 lim hole; do
    EC3D_LIB=$PWD/eddy_currents_3d_amd/libec3d_hip_base.so timeout -k 10 200 python3 tools/ab_perf.py $wl base >> $out/ab.log 2>> $out/ab.err
    timeout -k 10 200 python3 tools/ab_perf.py $wl int32-bookkeeping >> $out/ab.log 2>> $out/ab.err
  done
done
tail -n 3 $out/parity.log; cat $out/ab.log
