#!/bin/bash
# round 5, call s: the reference's shipped size (0.79 M unknowns, A-V) with fewer launches -- the fused kernels that already exist
# (K2 inside K3, K5 inside the next K1; no rendezvous involved), X every D-th iteration
out=gpurun_out/r05_s.log; : > $out
run() { echo "== $*" >> $out; env "$@" DICT_ONLY=1 timeout -k 10 120 python3 tools/quick_perf_av.py 1 1 1 >> $out 2>&1 || exit 1; }
run A=1
run EC3D_FUSE23=2
run EC3D_FUSE51=2
run EC3D_FUSE23=2 EC3D_FUSE51=2
run EC3D_FUSE23=2 EC3D_FUSE51=2 EC3D_XDEFER=4
run EC3D_XDEFER=4
run EC3D_XDEFER=2
run A=2
cat $out
