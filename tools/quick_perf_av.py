#!/usr/bin/env python3
"""On-box probe for the full A-V system: the shipped compare_to_Elmer geometry (tests/golden/g4),
optionally refined by an integer factor per axis (np.repeat keeps every material contiguous).
usage: quick_perf_av.py [fx fy fz] [dict=1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E

MU0 = 0.12566370964050292e-05
f = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else [1, 1, 1]
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_compare_to_Elmer.npz"))
vox = g["vox"]
vox = np.repeat(np.repeat(np.repeat(vox, f[2], axis=0), f[1], axis=1), f[0], axis=2)
sdz, sdy, sdx = vox.shape
dx = float(g["lattice_dim"])
flat = vox.reshape(-1)
ncell = flat.size
geo = flat.astype(np.int8).copy()
geo[geo == 0] = 6                      # one air domain is enough for the operator (D = 1 everywhere)
geoC = np.zeros(ncell, np.int32)
idx = np.flatnonzero(flat == 1)
if os.environ.get("AIR"):  # no conductor: the cost of the format without couplings
    idx = idx[:0]
geoC[idx] = 3 * ncell + 1 + np.arange(idx.size)
valPHYS = np.zeros((6, 5)); valPHYS[:, 0] = 1.0; valPHYS[0, 1] = MU0 * 35.26e6
b = np.zeros(3 * ncell + idx.size)
a = 183.0 / (6 * dx * 6 * dx)
b[np.flatnonzero(flat == 2)] = a * MU0; b[np.flatnonzero(flat == 3)] = -a * MU0
b[ncell + np.flatnonzero(flat == 4)] = a * MU0; b[ncell + np.flatnonzero(flat == 5)] = -a * MU0
for dic in (True,) if os.environ.get('DICT_ONLY') else (True, False):
    with E.EC3DSolver(dictionary=dic) as s:
        t = time.perf_counter()
        s.assemble(geo.reshape(sdz, sdy, sdx), geoC.reshape(sdz, sdy, sdx), valPHYS, np.full((3, 2), -0.95),
                   np.array([dx / f[0], dx / f[1], dx / f[2]]), 1e-3)
        ta = time.perf_counter() - t
        mi = s.info
        n = mi.n
        s.upload("B", b); s.upload("X", np.zeros(n))
        out = [f"grid {sdx}x{sdy}x{sdz} n={n} nnz={mi.nnz} tail_rows={mi.tail_rows} dict={mi.dict_classes} "
               f"nblk={s.geometry(0).nblk}/{s.geometry(1).nblk} zm_tpp={s.geometry(1).zm_tpp} ulist={s.geometry(0).ulist_n} assemble={ta * 1e3:.1f}ms"]
        for k in ("spmv", "k1", "k2", "k3", "k4", "k5"):
            try:
                out.append(f"{k}={s.time_kernel(k, 50) * 1e3:.1f}us")
            except E.EC3DError:
                out.append(f"{k}=fused")
        s.time_iterations(5)
        ms = s.time_iterations(50)
        out.append(f"iter={ms / 50 * 1e3:.1f}us -> {n * 50 / ms / 1e6:.2f} GDOF.it/s")
        t = time.perf_counter()
        x, it, _ = s.solve(b, np.zeros(n), 5e-3, 10000)
        wall = time.perf_counter() - t
        res = np.linalg.norm(b - s.spmv(x)) / np.linalg.norm(b)
        out.append(f"solve(tol 5e-3): iter={it} wall={1e3 * wall:.1f}ms true_resid={res:.2e} |x|={np.linalg.norm(x):.6e}")
        print(" ".join(out), flush=True)
