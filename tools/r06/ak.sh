#!/bin/bash
# round 6, call ak: the three-launch iteration at 20-30 Mi rows: the mid-size cache policy (next kernel's operand cacheable) or everything nontemporal?
out=$(pwd)/gpurun_out/r06ak; mkdir -p $out
for rep in 1 2; do
for g in 512x512x80 512x512x112; do
  timeout -k 10 300 python3 tools/ab_perf.py box:$g default_keep 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_KEEP=0 timeout -k 10 300 python3 tools/ab_perf.py box:$g keep=0 2>> $out/ab.err | cut -c1-150 | tee -a $out/ab.log
  EC3D_PLACE_VERBOSE=1 python3 - $g <<P 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $out/ab.log
import sys
sys.path.insert(0, ".")
import eddy_currents_3d_amd as E
g = tuple(int(a) for a in sys.argv[1].split("x"))
with E.EC3DSolver() as s:
    s.assemble_poisson(*g)
    print(g, "forced placement search:", [round(u, 1) for u in s.place_vectors(6)[0]])
P
done
done
