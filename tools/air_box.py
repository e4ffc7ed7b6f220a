#!/usr/bin/env python3
"""Scratch probe: the structured A-V form on a box of air (no conductor: three uncoupled 7-point components), per-kernel
averages inside the iteration.  usage: air_box.py sdx sdy sdz [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
sdx, sdy, sdz = (int(a) for a in sys.argv[1:4])
label = sys.argv[4] if len(sys.argv) > 4 else ""
ncell = sdx * sdy * sdz
geo = np.full((sdz, sdy, sdx), 6, np.int8)
geoC = np.zeros((sdz, sdy, sdx), np.int32)
valPHYS = np.zeros((6, 5)); valPHYS[:, 0] = 1.0
rng = np.random.Generator(np.random.PCG64(1))
b = rng.standard_normal(3 * ncell)
with E.EC3DSolver() as s:
    s.assemble(geo, geoC, valPHYS, np.full((3, 2), -0.95), np.array([1e-3] * 3), 1e-3)
    n = s.n
    s.upload("B", b); s.upload("X", np.zeros(n))
    s.iterate_begin(); s.iterate(1, 5); s.synchronize()
    K = 40
    ms = s.iterate(6, K, per_kernel=True); ms2 = s.iterate(6 + K, K, per_kernel=True)
    sp = s.time_kernel("spmv", 30)
    g = s.geometry(1)
    print(f"air {sdx}x{sdy}x{sdz} n={n} {label:12s} " + " ".join(f"{k}={1e3 * min(ms[k], ms2[k]):7.1f}" for k in ("k1", "k2", "k3", "k4", "k5")) +
          f" spmv={1e3 * sp:7.1f} us  wg={s.geometry(0).nblk}/{g.nblk} tpp={g.zm_tpp} pps={g.zm_pps}", flush=True)
