#!/usr/bin/env python3
"""Where does the wall time of ONE solve go at the reference's shipped size?  (compare_to_Elmer.vxc: 102 x 102 x 24 cells,
0.79 M unknowns, 173 iterations at the input's tolerance: 7.8 ms of kernels.)  The reference calls sprsBCGstabWR once per
time step (src/EC3D.f90:408) with b and x in host memory, so what surrounds the iterations -- the copies of b and x, the
set-up launches, the exit being noticed -- is paid per step.  Times, per part, over `reps` repetitions (median, ms):
upload of b / of x, the resident solve, download of x; the whole ec3d_solve; the F77 entry sprsbcgstabwr_ with the
reference's CSR arrays (matrix recognised from the previous call).
usage: solve_overhead.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
MU0 = 0.12566370964050292e-05
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g4_compare_to_Elmer.npz"))
vox = g["vox"]
sdz, sdy, sdx = vox.shape
dx = float(g["lattice_dim"])
flat = vox.reshape(-1)
ncell = flat.size
geo = flat.astype(np.int8).copy()
geo[geo == 0] = 6
geoC = np.zeros(ncell, np.int32)
idx = np.flatnonzero(flat == 1)
geoC[idx] = 3 * ncell + 1 + np.arange(idx.size)
valPHYS = np.zeros((6, 5)); valPHYS[:, 0] = 1.0; valPHYS[0, 1] = MU0 * 35.26e6
b = np.zeros(3 * ncell + idx.size)
a = 183.0 / (6 * dx * 6 * dx)
b[np.flatnonzero(flat == 2)] = a * MU0; b[np.flatnonzero(flat == 3)] = -a * MU0
b[ncell + np.flatnonzero(flat == 4)] = a * MU0; b[ncell + np.flatnonzero(flat == 5)] = -a * MU0
tol, itmax = 5e-3, 10000


def med(f, n=reps):
    ts = []
    out = None
    for _ in range(n):
        t = time.perf_counter()
        out = f()
        ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts)), out


with E.EC3DSolver() as s:
    s.assemble(geo.reshape(sdz, sdy, sdx), geoC.reshape(sdz, sdy, sdx), valPHYS, np.full((3, 2), -0.95),
               np.array([dx, dx, dx]), 1e-3)
    n = s.n
    x0 = np.zeros(n)
    s.solve(b, x0, tol, itmax)
    t_ub, _ = med(lambda: s.upload("B", b))
    t_ux, _ = med(lambda: s.upload("X", x0))

    def resident():
        s.upload("X", x0)
        t = time.perf_counter()
        it, _ = s.solve_resident(tol, itmax)
        return time.perf_counter() - t, it
    rs = [resident() for _ in range(reps)]
    t_res, it = 1e3 * float(np.median([r[0] for r in rs])), rs[0][1]
    t_dx, _ = med(lambda: s.download("X"))
    t_all, (x, it2, _) = med(lambda: s.solve(b, x0, tol, itmax))
    s.upload("B", b); s.upload("X", x0)
    s.iterate_begin(); s.iterate(1, 20); s.synchronize()
    t = time.perf_counter(); s.iterate(21, 100); s.synchronize()
    t_it = (time.perf_counter() - t) / 100 * 1e6
    valA, irow, jcol = s.export_csr()
print(f"n={n} iter={it}: kernels of one iteration {t_it:.1f} us -> {it * t_it * 1e-3:.2f} ms for the solve's iterations")
print(f"upload b {t_ub:.2f}  upload x {t_ux:.2f}  resident solve {t_res:.2f}  download x {t_dx:.2f}  | ec3d_solve {t_all:.2f} ms"
      f" (python wrapper included)")


def dropin():
    xx = x0.copy()
    t = time.perf_counter()
    k = E.sprsBCGstabWR(valA, irow, jcol, n, b, xx, tol, itmax)
    return time.perf_counter() - t, k, xx
first = dropin()
rs = [dropin() for _ in range(reps)]
print(f"sprsbcgstabwr_: first call {1e3 * first[0]:.1f} ms (matrix ingested), then {1e3 * float(np.median([r[0] for r in rs])):.2f} ms"
      f" per call, iter={rs[0][1]}, x equal to ec3d_solve's: {bool(np.array_equal(rs[0][2], x))}")
