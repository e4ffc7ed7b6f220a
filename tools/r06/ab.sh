#!/bin/bash
# round 6, call ab: does the vectors' placement matter to the plain-DIA five-launch iteration at 512^3 (SURVEY 8d's byte model)?
out=$(pwd)/gpurun_out/r06ab; mkdir -p $out
for i in 1 2 3; do
EC3D_PLACE_VERBOSE=1 timeout -k 10 300 python - <<P 2>&1 | grep -v amdgpu.ids | tee -a $out/dia.log
import numpy as np, sys
sys.path.insert(0, ".")
import eddy_currents_3d_amd as E, bench
N = 512
with E.EC3DSolver(dictionary=False) as s:
    s.assemble_poisson(N, N, N)
    print("bands:", s.band_placement())
    print("forced vector search:", s.place_vectors(6))
    s.upload("B", bench.bar_rhs(N)); s.upload("X", np.zeros(N ** 3))
    s.iterate_begin(); s.iterate(1, 4)
    ms = s.iterate(5, 20, per_kernel=True)
    print({k: round(v * 1e3, 1) for k, v in ms.items()}, "iter", round(s.time_iterations(50) / 50 * 1e3, 1))
P
done
