#!/usr/bin/env python3
"""On-box probe: per-kernel time of the single-component operator on an sdx x sdy x sdz box.
usage: quick_perf_box.py sdx sdy sdz   (EC3D_ZMARCH=0/1 etc. from the environment)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E

sdx, sdy, sdz = (int(a) for a in sys.argv[1:4])
with E.EC3DSolver() as s:
    s.assemble_poisson(sdx, sdy, sdz)
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(1))
    s.upload("B", rng.standard_normal(n)); s.upload("X", np.zeros(n))
    g0, g1 = s.geometry(0), s.geometry(1)
    out = [f"{sdx}x{sdy}x{sdz} n={n} nblk={g0.nblk}/{g1.nblk} zm_tpp={g1.zm_tpp}"]
    for k in ("spmv", "k1", "k2", "k3", "k4", "k5"):
        out.append(f"{k}={s.time_kernel(k, 50) * 1e3:.1f}us")
    s.time_iterations(5)
    ms = s.time_iterations(50)
    out.append(f"iter={ms / 50 * 1e3:.1f}us -> {n * 50 / ms / 1e6:.2f} GDOF.it/s")
    print(" ".join(out), flush=True)
