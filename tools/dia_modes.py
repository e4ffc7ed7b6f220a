#!/usr/bin/env python3
"""Scratch probe: plain-DIA SpMV at 512^3, several handles in ONE process, with / without the placement probe
(EC3D_PLACE candidates; EC3D_PLACE_VERBOSE prints every candidate's time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
N = 512
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    with E.EC3DSolver(dictionary=False) as s:
        s.assemble_poisson(N, N, N)
        s.upload("X", np.zeros(N ** 3))
        t = [s.time_kernel("spmv", 20) * 1e3 for _ in range(2)]
    print(f"trial {trial} EC3D_PLACE={os.environ.get('EC3D_PLACE', 'default')}: spmv {min(t):.1f} us", flush=True)
