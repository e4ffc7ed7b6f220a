#!/bin/bash
# round 6, call ah: where the placement search's time goes on large grids (fresh process per grid)
out=$(pwd)/gpurun_out/r06ah; mkdir -p $out
for g in "512 512 512" "640 640 640" "768 768 768"; do
EC3D_PLACE_VERBOSE=1 timeout -k 10 600 python - $g <<P 2>&1 | grep -v amdgpu.ids | tee -a $out/grids.log
import numpy as np, sys, time
sys.path.insert(0, ".")
import eddy_currents_3d_amd as E
g = tuple(int(a) for a in sys.argv[1:4])
with E.EC3DSolver() as s:
    t0 = time.perf_counter(); s.assemble_poisson(*g); ta = time.perf_counter() - t0
    us, kept, ms = s.vector_placement()
    print(g, "candidates", [round(u, 1) for u in us], "kept", kept, f"search {ms:.0f} ms of {ta * 1e3:.0f} ms assembly", flush=True)
P
done
