#!/bin/bash
# round 6, call ag: the placement search on other grids of the three-launch iteration (what it sees, what it costs)
out=$(pwd)/gpurun_out/r06ag; mkdir -p $out
timeout -k 10 600 python - <<P 2>&1 | grep -v amdgpu.ids | tee $out/grids.log
import numpy as np, sys, time
sys.path.insert(0, ".")
import eddy_currents_3d_amd as E
for g in ((512, 512, 128), (384, 384, 384), (512, 512, 256), (640, 640, 640), (768, 768, 768)):
    n = g[0] * g[1] * g[2]
    with E.EC3DSolver() as s:
        t0 = time.perf_counter(); s.assemble_poisson(*g); ta = time.perf_counter() - t0
        us, kept, ms = s.vector_placement()
        s.upload("B", np.ones(n)); s.upload("X", np.zeros(n))
        s.iterate_begin(); s.iterate(1, 4)
        it = s.time_iterations(40) / 40 * 1e3
        print(f"{g} n={n/1e6:.1f} M: fusion {s.fusion()} candidates {[round(u, 1) for u in us]} kept {kept} search {ms:.0f} ms of {ta * 1e3:.0f} ms assembly; "
              f"iteration {it:.1f} us = {n / it / 1e3:.2f} G DOF.it/s", flush=True)
P
