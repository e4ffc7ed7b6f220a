#!/bin/bash
# round 6, call ao: do planes on 2 MiB boundaries (ghost zones of a whole number of planes) change the allocations' levels?
out=$(pwd)/gpurun_out/r06ao; mkdir -p $out
EC3D_PLACE_VEC=0 timeout -k 10 400 python tools/vec_place_probe.py 5 2>&1 | grep -v amdgpu.ids | grep "round 0\|round 2" | tee $out/default.log
EC3D_EXP_GHOST_ALIGN=262144 EC3D_PLACE_VEC=0 timeout -k 10 400 python tools/vec_place_probe.py 5 2>&1 | grep -v amdgpu.ids | grep "round 0\|round 2" | tee $out/aligned.log
