#!/usr/bin/env python3
"""Scratch probe (round 6): does the three-launch iteration's time at 512^3 depend on WHERE the vectors lie?  Several
handles in ONE process, every one kept alive while the next allocates (so that each lands on memory of its own), the
iteration's kernels timed on each -- three times over, to tell a property of the allocation from noise.  With
EC3D_PLACE_VEC=0 the handles take what the driver gives them; otherwise each reports what its placement probe saw."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
import bench
N = 512
H = int(sys.argv[1]) if len(sys.argv) > 1 else 5
b = bench.bar_rhs(N)
hs = []
for h in range(H):
    s = E.EC3DSolver()
    s.assemble_poisson(N, N, N)
    us, kept, ms = s.vector_placement()
    print(f"handle {h}: placement candidates {[round(u, 1) for u in us]} us, kept {kept}, search {ms:.0f} ms", flush=True)
    s.upload("B", b)
    s.upload("X", np.zeros(N ** 3))
    hs.append(s)
for rnd in range(3):
    for h, s in enumerate(hs):
        s.iterate_begin()
        s.iterate(1, 4)
        ms = s.iterate(5, 24, per_kernel=True)
        itr = s.time_iterations(100) / 100
        print(f"round {rnd} handle {h}: " + " ".join(f"{k}={1e3 * v:7.1f}" for k, v in ms.items()) + f"  iter={1e3 * itr:7.1f} us", flush=True)
for s in hs:
    s.close()
