#!/bin/bash
# round 6, call am: where the interleaved A-V march's extra fetches come from: il0 as shipped; il1 without the one-sided face slots'
# loads; il2 without the U block's +-sdx / edge loads; il3 without U tiles at all (results wrong in il1-il3: counters and times only)
out=$(pwd)/gpurun_out/r06am; mkdir -p $out; REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in il0 il1 il2 il3; do
  for ctr in FETCH_SIZE; do
    EC3D_LIB=$REPO/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$v -- python3 $REPO/tools/av256_perf.py $v > $out/$v.log 2> $out/$v.err || { echo "$v failed"; tail -5 $out/$v.err; }
    echo "== $v $ctr"; python3 $REPO/tools/pmc_avg.py $out/$v spmv | tee -a $out/summary.log
  done
  rm -rf $out/$v
  EC3D_LIB=$REPO/tools/abtmp/libec3d_hip_$v.so timeout -k 10 300 python3 $REPO/tools/av256_perf.py $v 2>/dev/null | tail -n 1 | cut -c1-12,150-330 | tee -a $out/summary.log
done
