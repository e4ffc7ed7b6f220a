"""Shipped-size parity (BASELINE config 1): the reference's own compare_to_Elmer.vxc geometry
(102x102x24, n = 792 288, nnz = 5 892 072), step 0, against numbers captured from the unmodified
reference (tests/golden/g4_*.npz: iter, ||b||, ||x||, 200 probes of x, row-length histogram)."""
import time

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
MU0 = 0.12566370964050292e-05  # src/vxc2data.f90:402


def step0_rhs(vox, dx):
    """src/EC3D.f90:245-256, :345-365 at t = 0 with Uaf = 0: Jaf = mu0*F on the coil cells
    (F = +-183/(6*dx*6*dz), materials 2..5 = axp, axm, ayp, aym)."""
    ncell = vox.size
    flat = vox.reshape(-1)
    a = 183.0 / (6 * dx * 6 * dx)
    b = np.zeros(3 * ncell + int((flat == 1).sum()))
    b[np.flatnonzero(flat == 2)] = a * MU0
    b[np.flatnonzero(flat == 3)] = -a * MU0
    b[ncell + np.flatnonzero(flat == 4)] = a * MU0
    b[ncell + np.flatnonzero(flat == 5)] = -a * MU0
    return b


def test_compare_to_elmer_step0(oracle):
    import eddy_currents_3d_amd as E
    from oracle import make_goldens as G
    g = load_golden("g4_compare_to_Elmer")
    vox = g["vox"]
    dx = float(g["lattice_dim"])
    geo, geoC, _ = G.geometry_tables(vox, [1], 5)
    valPHYS = np.zeros((int(geo.max()), 5)); valPHYS[:, 0] = 1.0
    valPHYS[0, 1] = MU0 * 35.26e6
    b = step0_rhs(vox, dx)
    assert np.linalg.norm(b) == pytest.approx(float(g["bnorm"][0]), rel=1e-15)
    tol = float(g["tol"])
    with E.EC3DSolver() as s:
        t0 = time.perf_counter()
        s.assemble(geo, geoC, valPHYS, np.full((3, 2), -0.95), np.full(3, dx), 1e-3)
        t_asm = time.perf_counter() - t0
        assert s.n == int(g["n"]) and s.info.nnz == int(g["nnz"])
        va, ir, jc = s.export_csr()
        assert np.array_equal(np.bincount(np.diff(ir), minlength=14), g["rowlen_hist"])
        # the oracle's restatement of the assembly gives the same CSR (bitwise)
        m = oracle.gen_sparse_matrix(geo, geoC, valPHYS, np.full((3, 2), -0.95), np.full(3, dx), 1e-3)
        assert np.array_equal(m["irow"], ir) and np.array_equal(m["jcol"], jc) and np.array_equal(m["valA"], va)
        t0 = time.perf_counter()
        x, it, _ = s.solve(b, np.zeros_like(b), tol, 10000)
        t_solve = time.perf_counter() - t0
    it_ref = int(g["iters"][0])
    print(f"compare_to_Elmer step 0: iter gpu {it} / reference {it_ref}; ||x|| {np.linalg.norm(x):.6e} / "
          f"{float(g['xnorm'][0]):.6e}; assemble {t_asm * 1e3:.1f} ms, solve {t_solve * 1e3:.1f} ms "
          f"(reference: {float(g['seconds'][0]):.2f} s on one core here)")
    assert abs(it - it_ref) <= 0.15 * it_ref
    assert np.linalg.norm(x) == pytest.approx(float(g["xnorm"][0]), rel=10 * tol)
    assert np.abs(x[g["probes"]] - g["xprobe"][0]).max() <= 10 * tol * np.abs(x).max()
