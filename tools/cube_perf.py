#!/usr/bin/env python3
"""On-box probe: the N^3 cube (dictionary format) -- bare SpMV / K1 / K3 times under the environment's knobs.  cube_perf.py [label] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import eddy_currents_3d_amd as E
label = sys.argv[1] if len(sys.argv) > 1 else "cube"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with E.EC3DSolver() as s:
    s.assemble_poisson(N, N, N)
    s.upload("B", bench.bar_rhs(N)); s.upload("X", np.zeros(N ** 3))
    g = s.geometry(1)
    out = [f"[{label}] {N}^3 nblk={g.nblk} patch={g.patch_x}x{g.patch_y} fusion={s.fusion()}"]
    for k in ("spmv", "k1", "k3"):
        try:
            out.append(f"{k}={s.time_kernel(k, 30) * 1e3:.1f}us")
        except E.EC3DError:
            out.append(f"{k}=fused")
    print(" ".join(out), flush=True)
