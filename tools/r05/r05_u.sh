#!/bin/bash
# round 5, call u: the shipped size on runtime-shaped 2-D tiles (EC3D_SAV_PATCH), where K2-in-K3 / K5-in-K1 exist for the A-V form
out=gpurun_out/r05_u.log; : > $out
run() { echo "== $*" >> $out; env "$@" DICT_ONLY=1 timeout -k 10 120 python3 tools/quick_perf_av.py 1 1 1 >> $out 2>&1 || exit 1; }
run A=1
run EC3D_SAV_PATCH=2
run EC3D_SAV_PATCH=2 EC3D_FUSE23=2
run EC3D_SAV_PATCH=2 EC3D_FUSE51=2
run EC3D_SAV_PATCH=2 EC3D_FUSE23=2 EC3D_FUSE51=2
run EC3D_SAV_PATCH=2 EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_XDEFER=4
grep -v amdgpu.ids $out
