#!/bin/bash
# round 6, call a: where BASELINE config 3 at 256^3 stands on round 5's build (knob survey + rocprof/PMC baseline)
out=gpurun_out/r06a; mkdir -p $out
run() { local name=$1; shift; echo "== $name"; timeout -k 10 240 env "$@" python tools/av256_perf.py $name >> $out/perf.log 2>> $out/perf.err; echo "rc=$?"; tail -n 1 $out/perf.log; }
run default X=1 || exit 1
run patch EC3D_SAV_PATCH=1 || exit 1
run patch_fused EC3D_SAV_PATCH=1 EC3D_FUSE23=2 EC3D_FUSE51=2 || exit 1
run nblk1536 EC3D_NBLK_SPMV=1536 || exit 1
run nblk2048 EC3D_NBLK_SPMV=2048 || exit 1
run nblk768 EC3D_NBLK_SPMV=768 || exit 1
timeout -k 10 500 bash tools/profile_bench.sh r06_av256_base 256 dict av256 > $out/profile.log 2>&1; echo "profile rc=$?"
tail -n 30 $out/profile.log
