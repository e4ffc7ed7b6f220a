#!/usr/bin/env python3
"""Scratch probe: do the vector kernels' times at 512^3 depend on where the driver puts the vectors?  Several handles in
ONE process (each allocates its vectors anew), K2 / K4 / K5 timed in isolation, optional ballast in between."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eddy_currents_3d_amd as E
N = 512
ballast = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if trial:
        ballast.append(torch.empty((trial * 53 + 7) << 20, dtype=torch.uint8, device="cuda"))
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        s.upload("X", np.zeros(N ** 3))
        s.upload("B", np.ones(N ** 3))
        s.iterate_begin(); s.iterate(1, 3); s.synchronize()
        ms = s.iterate(4, 20, per_kernel=True)
        t = {k: min(s.time_kernel(k, 20) for _ in range(2)) * 1e3 for k in ("k2", "k4", "k5", "spmv")}
    print(f"trial {trial}: isolated " + " ".join(f"{k}={v:7.1f}" for k, v in t.items()) + "   in the iteration " +
          " ".join(f"{k}={1e3 * v:7.1f}" for k, v in ms.items()), flush=True)
