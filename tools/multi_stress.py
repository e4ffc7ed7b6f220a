#!/usr/bin/env python3
"""Determinism stress of the in-library multi-GPU loop on one card: the same right-hand sides solved again and again
on 3, 4 and 8 slabs must give the same bits every time (a halo plane read too early, or a partial sum read before
its producer finished, shows as a difference or as the watchdog's report).  python tools/multi_stress.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import eddy_currents_3d_amd as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g3_moving_coil_18x16x12.npz"))
rng = np.random.Generator(np.random.PCG64(4))
bad = 0
for world in (3, 4, 8):
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble_poisson(64, 64, 48)
        bs = [rng.standard_normal(m.n) for _ in range(3)]
        ref = [m.solve(b, np.zeros(m.n), 1e-8, 5000) for b in bs]
        for r in range(reps):
            for b, (xr, itr) in zip(bs, ref):
                x, it = m.solve(b, np.zeros(m.n), 1e-8, 5000)
                bad += int(it != itr or not np.array_equal(x, xr))
    if world > 4:      # the 12 planes of the A-V fixture do not cut 8 ways (two planes per rank at least)
        print(f"{world} slabs: {reps} repetitions of 3 cube solves, differences so far: {bad}", flush=True)
        continue
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        ref = [m.solve(g[f"b{k}"], g[f"xin{k}"], float(g["tol"]), int(g["itmax"])) for k in range(4)]
        for r in range(reps):
            for k, (xr, itr) in enumerate(ref):
                x, it = m.solve(g[f"b{k}"], g[f"xin{k}"], float(g["tol"]), int(g["itmax"]))
                bad += int(it != itr or not np.array_equal(x, xr))
    print(f"{world} slabs: {reps} repetitions of 3 cube solves and 4 A-V solves, differences so far: {bad}", flush=True)
sys.exit(1 if bad else 0)
