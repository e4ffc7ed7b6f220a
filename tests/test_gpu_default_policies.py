"""The kernel instances the library selects BY ITSELF at the sizes the metric is quoted on, against the oracle's
GPU-order twin, bit for bit.

choose_sweep (csrc/ec3d_context.hip) switches by size: nontemporal streams and the X update every fourth iteration from
4.5 Mi rows, the next kernel's operand kept cacheable up to 32 Mi rows, and from 20 Mi rows (undivided handle; z-slabs: 32 Mi) on 2-D tiles the three-launch iteration (K2 inside K3, K5 inside
the next K1, P / AP in alternating buffers, K4 as an SpMV kernel that computes A S again) with the X update applied
every fourth iteration (six iterations = one whole group and the itmax exit's partial one).  The small
parity cases force those instances through EC3D_NT / EC3D_KEEP / EC3D_FUSE* (tests/test_gpu_parity.py); here NOTHING is
forced -- the handle is built the way bench.py builds it and the twin follows the launch geometry the library reports
(ec3d_get_visit_order), for the first iterations of src/solvers.f90:24-50 (the itmax exit of :25-29 ends the run; the
twin costs seconds per iteration on one host core at these sizes).  x, iter and every ||S||, ||R|| of the history must
be the twin's, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


def no_knobs(monkeypatch):
    for k in ("EC3D_NT", "EC3D_KEEP", "EC3D_FUSE23", "EC3D_FUSE51", "EC3D_PATCH", "EC3D_NBLK", "EC3D_NBLK_SPMV",
              "EC3D_VEC_DEPTH", "EC3D_XCD_MAP", "EC3D_PITCH", "EC3D_ZMARCH", "EC3D_SAV_PATCH", "EC3D_XDEFER",
              "EC3D_XD_OFF_DEPTH", "EC3D_XD_ON_DEPTH", "EC3D_K4S"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("dims, fused, iters", [((256, 256, 80), False, 8), ((512, 512, 72), False, 6), ((512, 512, 80), True, 6),
                                                ((512, 512, 128), True, 6), ((512, 512, 256), True, 6)],
                         ids=["5Mi-rows-nt-keep", "18Mi-rows-five-launches", "20Mi-rows-three-launches", "32Mi-rows-three-launches",
                              "64Mi-rows-three-launches"])
def test_cube_at_the_default_policy_bitwise(E, oracle, dims, fused, iters, monkeypatch):
    """Single-component operator (BASELINE configs 2 / 4 family).  256 x 256 x 80 = 5.2 M rows: nontemporal streams,
    AP / S / R kept cacheable, 2-D tiles, five launches.  512 x 512 x 72 = 18 Mi rows: the largest five-launch size
    class.  512 x 512 x 80 = 20 Mi rows: the size from which, on an undivided handle, K2 runs inside K3, K5 inside the next
    K1, K4 as an SpMV kernel that computes A S again (k4s_x_r_spmv) and X is updated every fourth iteration by a launch of
    its own, all by themselves (32 Mi rows until round 6) -- the configuration of the headline 512^3 run; 512 x 512 x 128 (from
    where the work vectors' placement is searched too) and 512 x 512 x 256 the same on more planes."""
    no_knobs(monkeypatch)
    sdx, sdy, sdz = dims
    n = sdx * sdy * sdz
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(4096))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    with E.EC3DSolver() as s:
        s.assemble_poisson(sdx, sdy, sdz)
        g1, g2 = s.geometry(1), s.geometry(2)
        assert g1.patch_x == 128 and g1.zm_tpp == sdx * sdy // 512
        assert (g2.patch_x == 128 and g2.nblk == g1.nblk) == fused      # S.S summed inside the SpMV kernel when fused
        assert s.fusion() == ((1, 1) if fused else (0, 0))
        assert s.x_interval() == 4                 # X updated every 4th iteration from 4.5 Mi rows (k4d_x_r_update / k4s)
        g0 = s.geometry(0)                                        # fused: K4 runs as an SpMV kernel (k4s_x_r_spmv)
        assert (g0.patch_x == 128 and g0.nblk == g1.nblk) == fused
        x, it, hist = s.solve(b, x0, 1e-30, iters - 1, hist_cap=iters)
        xo, ito, hs, hr = oracle.twin_solve(s, valA, irow, jcol, b, x0, 1e-30, iters - 1, hist_cap=iters)
    assert it == ito == iters
    assert np.array_equal(hist[:iters, 0], hs[:iters]) and np.array_equal(hist[:iters, 1], hr[:iters])
    assert np.array_equal(x, xo)
    print(f"{sdx}x{sdy}x{sdz}: {iters} iterations at the default policy bit-identical to the twin "
          f"(||R|| {hist[0, 1]:.6e} -> {hist[iters - 1, 1]:.6e})")


def test_av_system_at_the_default_policy_bitwise(E, oracle, monkeypatch):
    """The reference's own system [Ax | Ay | Az | U] (src/EC3D.f90:408) at a size where the structured form runs
    pitched, z-marching and with nontemporal streams by itself: the shipped compare_to_Elmer geometry refined x2 per
    axis (204 x 204 x 48, n = 6.3 M; 8 M device rows).  The matrix is the oracle's restatement of gen_sparse_matrix
    (src/EC3D.f90:465-1049), which the device assembly must reproduce entry for entry."""
    no_knobs(monkeypatch)
    from bench import av_system
    geo, geoC, valPHYS, BND, delta, dt, b = av_system(2)
    m = oracle.gen_sparse_matrix(geo, geoC, valPHYS, BND, delta, dt)
    n = m["n"]
    assert n == len(b)
    iters = 8
    x0 = np.zeros(n)
    with E.EC3DSolver() as s:
        s.assemble(geo, geoC, valPHYS, BND, delta, dt)
        assert s.info.tail_rows == 0 and s.geometry(1).zm_tpp > 0           # structured, pitched, z-marching
        assert s.x_interval() == 4 and s.fusion() == (0, 0)                 # five launches, X every fourth iteration
        va, ir, jc = s.export_csr()
        assert np.array_equal(ir, m["irow"]) and np.array_equal(jc, m["jcol"]) and np.array_equal(va, m["valA"])
        x, it, hist = s.solve(b, x0, 1e-30, iters - 1, hist_cap=iters)
        xo, ito, hs, hr = oracle.twin_solve(s, m["valA"], m["irow"], m["jcol"], b, x0, 1e-30, iters - 1, hist_cap=iters)
    assert it == ito == iters
    assert np.array_equal(hist[:iters, 0], hs[:iters]) and np.array_equal(hist[:iters, 1], hr[:iters])
    assert np.array_equal(x, xo)
    print(f"A-V 204x204x48 (n = {n}): {iters} iterations at the default policy bit-identical to the twin")
