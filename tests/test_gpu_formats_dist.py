"""GPU tests (1 GPU) for the two band storage formats and for the z-slab building blocks."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


@pytest.mark.parametrize("name", ["g2_conducting_hole_16x15x14", "g2v_conducting_moving_16x15x14",
                                  "g3_moving_coil_18x16x12"])
def test_dictionary_and_plain_dia_are_bit_identical(E, oracle, name):
    """Dictionary form (1 class byte/row + table) multiplies the same doubles in the same order."""
    g = load_golden(name)
    n = len(g["irow"]) - 1
    x = np.random.Generator(np.random.PCG64(4)).standard_normal(n)
    ref = oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x)
    res = {}
    for route in ("csr", "assemble"):
        for dic in (False, True):
            with E.EC3DSolver(dictionary=dic, structured=False) as s:
                if route == "csr":
                    s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
                else:
                    s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
                assert (s.info.dict_classes > 0) == dic
                assert np.array_equal(s.spmv(x), ref)
                va, ir, jc = s.export_csr()
                assert np.array_equal(ir, g["irow"]) and np.array_equal(jc, g["jcol"]) and np.array_equal(va, g["valA"])
                res[(route, dic)] = s.solve(g["b0"], g["xin0"], float(g["tol"]), int(g["itmax"]), hist_cap=64)
    x0, it0, h0 = res[("csr", False)]
    for k, (xk, itk, hk) in res.items():
        assert itk == it0 and np.array_equal(xk, x0) and np.array_equal(hk[:itk], h0[:itk]), k


def test_dictionary_falls_back_when_too_many_classes(E, oracle):
    """More than 256 distinct coefficient tuples: plain DIA streams are kept, results unchanged."""
    N = 10
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    valA = valA * (1.0 + 1e-3 * np.arange(len(valA)))  # every row different
    x = np.random.Generator(np.random.PCG64(8)).standard_normal(N ** 3)
    with E.EC3DSolver(dictionary=True) as s:
        s.set_matrix_csr(valA, irow, jcol)
        assert s.info.dict_classes == 0 and s.info.nbands == 7
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))


@pytest.mark.parametrize("dic", [False, True])
def test_poisson_formats_match_oracle(E, oracle, dic):
    N = 40
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    x = np.random.Generator(np.random.PCG64(6)).standard_normal(N ** 3)
    with E.EC3DSolver(dictionary=dic) as s:
        s.assemble_poisson(N, N, N)
        assert (s.info.dict_classes > 0) == dic
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))
        va, ir, jc = s.export_csr()
        assert np.array_equal(ir, irow) and np.array_equal(jc, jcol) and np.array_equal(va, valA)


# ------------------------------------------------------------------------------------- slabs
def test_single_slab_dist_path_equals_single_gpu_solve_bitwise(E, oracle):
    """world = 1 through ec3d_dist_step (finalize -> lsum -> gsum) == ec3d_solve, bit for bit."""
    from eddy_currents_3d_amd.dist import SlabSolver
    N, tol = 24, 1e-8
    b = oracle.bar_rhs(N)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x_ref, it_ref, _ = s.solve(b, np.zeros(N ** 3), tol, 10000)
    sl = SlabSolver.poisson_cube(N, 0, 1)
    sl.set_rhs(b, np.zeros(N ** 3))
    it = sl.solve(tol, 10000, poll=5)
    assert it == it_ref
    assert np.array_equal(sl.gather_x(), x_ref)


@pytest.mark.parametrize("world,N", [(2, 24), (3, 20), (4, 32)])
def test_slabs_on_one_gpu_match_single_gpu(E, oracle, world, N):
    """`world` z-slabs held by one process on one GPU (InProcessSlabs: same schedule, halo planes copied
    tensor to tensor, per-slab sums concatenated) vs the undivided solve."""
    from eddy_currents_3d_amd.dist import HipSlabOps, InProcessSlabs, slab_bounds
    tol = 1e-8
    b = oracle.bar_rhs(N)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x_ref, it_ref, _ = s.solve(b, np.zeros(N ** 3), tol, 10000)
    ops = []
    for r in range(world):
        k0, k1 = slab_bounds(N, r, world)
        o = HipSlabOps(N, N, N, k0, k1, world)
        o.set_vector("B", b.reshape(N, N * N)[k0:k1].reshape(-1))
        ops.append(o)
    drv = InProcessSlabs(ops)
    it = drv.solve(tol, 10000)
    x = drv.x()
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    res = np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b)
    print(f"{world} slabs of {N}^3: iter {it} / undivided {it_ref}, true residual {res:.2e}, "
          f"rel diff {np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref):.2e}")
    assert res < 5 * tol
    assert np.linalg.norm(x - x_ref) <= 1e-5 * np.linalg.norm(x_ref)
    assert abs(it - it_ref) <= 0.35 * it_ref


def test_slab_spmv_equals_rows_of_global_operator(E, oracle):
    """A slab's operator applied to [lower ghost | owned | upper ghost] == the matching rows of A x."""
    from eddy_currents_3d_amd.dist import HipSlabOps, K1
    N, k0, k1 = 18, 5, 11
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    x = np.random.Generator(np.random.PCG64(10)).standard_normal(N ** 3)
    y = oracle.spmv_csr(valA, irow, jcol, x)
    kdz = N * N
    o = HipSlabOps(N, N, N, k0, k1, 1)
    o.set_vector("P", x[k0 * kdz:k1 * kdz])
    lo_send, lo_recv, hi_send, hi_recv = o.halo_views("P")
    import torch
    with o.context():
        lo_recv.copy_(torch.from_numpy(x[(k0 - 1) * kdz:k0 * kdz].copy()))
        hi_recv.copy_(torch.from_numpy(x[k1 * kdz:(k1 + 1) * kdz].copy()))
        o.step(K1, 1)
    o.synchronize()
    assert np.array_equal(o.get_vector("AP"), y[k0 * kdz:k1 * kdz])


def test_slab_handle_refuses_whole_solve(E):
    with E.EC3DSolver() as s:
        s.assemble_poisson(16, 16, 16, slab=(4, 8))
        with pytest.raises(E.EC3DError, match="z-slab"):
            s.solve(np.zeros(4 * 256), np.zeros(4 * 256), 1e-6, 10)


# ------------------------------------------------------------------------------ z-marching map
@pytest.mark.parametrize("dims", [(64, 64, 24), (128, 32, 16)])
@pytest.mark.parametrize("dic", [False, True])
def test_zmarch_spmv_and_solve_bitwise(E, oracle, dims, dic):
    """Grids whose xy-plane is a whole number of 512-row tiles use the z-marching SpMV map (x of the
    planes below/at the row carried in registers).  Same products, same row-sum order: SpMV bit-identical
    to the oracle; the solve bit-identical to the oracle's twin run with the two launch geometries."""
    sdx, sdy, sdz = dims
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(21))
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    with E.EC3DSolver(dictionary=dic) as s:
        s.assemble_poisson(sdx, sdy, sdz)
        gs = s.geometry(1)
        assert gs.zm_tpp == sdx * sdy // 512 and gs.zm_pps >= 2
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))
        xs, it, hist = s.solve(b, np.zeros(n), 1e-9, 5000, hist_cap=64)
        xo, ito, hs, hr = oracle.twin_solve(s, valA, irow, jcol, b, np.zeros(n), 1e-9,
                                                      5000, hist_cap=64)
        assert it == ito and np.array_equal(xs, xo)
        k = min(it, 64)
        assert np.array_equal(hist[:k, 0], hs[:k])
        s.set_zmarch(False)
        assert s.geometry(1).zm_tpp == 0
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))
        x2, it2, _ = s.solve(b, np.zeros(n), 1e-9, 5000)
        xo2, ito2, _, _ = oracle.twin_solve(s, valA, irow, jcol, b, np.zeros(n), 1e-9, 5000)
        assert it2 == ito2 and np.array_equal(x2, xo2)


def test_zmarch_slabs_on_one_gpu(E, oracle):
    from eddy_currents_3d_amd.dist import HipSlabOps, InProcessSlabs, slab_bounds
    sdx, sdy, sdz, world, tol = 64, 64, 32, 2, 1e-8
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    b = np.random.Generator(np.random.PCG64(5)).standard_normal(sdx * sdy * sdz)
    ops = []
    for r in range(world):
        k0, k1 = slab_bounds(sdz, r, world)
        o = HipSlabOps(sdx, sdy, sdz, k0, k1, world)
        assert o.local.geometry(1).zm_tpp == 8
        o.set_vector("B", b.reshape(sdz, sdx * sdy)[k0:k1].reshape(-1))
        ops.append(o)
    drv = InProcessSlabs(ops)
    it = drv.solve(tol, 5000)
    x = drv.x()
    assert np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b) < 5 * tol


# --------------------------------------------------------------------------- A-V system in slabs
@pytest.mark.parametrize("name,world", [("g2_conducting_hole_16x15x14", 2), ("g2_conducting_hole_16x15x14", 3),
                                        ("g3_moving_coil_18x16x12", 2), ("g2v_conducting_moving_16x15x14", 4)])
@pytest.mark.parametrize("structured", [True, False])
@pytest.mark.parametrize("vsplit", [False, True])
def test_av_slabs_on_one_gpu_match_reference(E, name, world, structured, plane_pitch, vsplit):
    """The full A-V system [Ax|Ay|Az|U] cut into z-slabs (extended grid: 2 halo planes per side, inert halo
    rows, ownership-masked dot products), all slabs held by one process on one GPU; the cuts go through
    the conductor.  Against the unmodified reference's solution of the same captured system."""
    from eddy_currents_3d_amd.dist import HipAVSlabOps, InProcessSlabs, slab_bounds
    g = load_golden(name)
    sdz = g["geoPHYS"].shape[0]
    n = len(g["irow"]) - 1
    tol, itmax = float(g["tol"]), int(g["itmax"])
    for k in (0, 1):
        ops = []
        for r in range(world):
            k0, k1 = slab_bounds(sdz, r, world)
            o = HipAVSlabOps(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]),
                             k0, k1, world, structured=structured)
            assert o.structured == structured
            o.set_vector_global("B", g[f"b{k}"])
            o.set_vector_global("X", g[f"xin{k}"])
            ops.append(o)
        drv = InProcessSlabs(ops, vsplit=vsplit)   # vsplit: K2/K5 boundary tiles first (exchange hidden)
        it = drv.solve(tol, itmax)
        x = drv.x(n)
        xr = g[f"xout{k}"]
        rel = np.linalg.norm(x - xr) / np.linalg.norm(xr)
        print(f"{name} in {world} slabs, step {k}: iter {it} / reference {int(g['iters'][k])}, rel diff {rel:.2e}")
        assert rel <= 10 * tol
        assert abs(it - int(g["iters"][k])) <= max(3, 0.15 * int(g["iters"][k]))


@pytest.mark.parametrize("world,sdz", [(2, 40), (3, 48)])
def test_overlap_split_launches_match_unsplit(E, oracle, world, sdz):
    """K1/K3 as interior + boundary launches (the schedule that hides the halo exchange) vs the plain
    schedule on the same slabs: same converged solution; interior launches do not read the ghost planes
    (the exchange happens between the two launches)."""
    from eddy_currents_3d_amd.dist import HipSlabOps, InProcessSlabs, slab_bounds
    sdx = sdy = 64
    tol = 1e-9
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    b = np.random.Generator(np.random.PCG64(3)).standard_normal(sdx * sdy * sdz)
    res = {}
    for overlap in (True, False):
        ops = []
        for r in range(world):
            k0, k1 = slab_bounds(sdz, r, world)
            o = HipSlabOps(sdx, sdy, sdz, k0, k1, world)
            assert o.can_overlap()
            o.set_vector("B", b.reshape(sdz, sdx * sdy)[k0:k1].reshape(-1))
            ops.append(o)
        drv = InProcessSlabs(ops, overlap=overlap)
        it = drv.solve(tol, 5000)
        x = drv.x()
        assert np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b) < 5 * tol
        res[overlap] = (it, x)
        for o in ops:
            o.close()
    print(f"{world} slabs of {sdx}x{sdy}x{sdz}: iterations split {res[True][0]} / unsplit {res[False][0]}")
    assert np.linalg.norm(res[True][1] - res[False][1]) <= 1e-6 * np.linalg.norm(res[False][1])


def test_split_spmv_is_the_same_operator(E, oracle):
    """AP from K1_INT + K1_BND == AP from K1, bit for bit (same row sums, only the launch geometry differs)."""
    from eddy_currents_3d_amd.dist import HipSlabOps, K1, K1_BND, K1_INT
    sdx, sdy, sdz, k0, k1 = 64, 64, 40, 8, 28
    kdz = sdx * sdy
    rng = np.random.Generator(np.random.PCG64(12))
    x = rng.standard_normal(sdx * sdy * sdz)
    import torch
    out = []
    for stages in ((K1,), (K1_INT, K1_BND)):
        o = HipSlabOps(sdx, sdy, sdz, k0, k1, 1)
        o.set_vector("P", x[k0 * kdz:k1 * kdz])
        lo_s, lo_r, hi_s, hi_r = o.halo_views("P")
        with o.context():
            lo_r.copy_(torch.from_numpy(x[(k0 - 1) * kdz:k0 * kdz].copy()))
            hi_r.copy_(torch.from_numpy(x[k1 * kdz:(k1 + 1) * kdz].copy()))
            for st in stages:
                o.step(st, 1)
        o.synchronize()
        out.append(o.get_vector("AP"))
        o.close()
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    y = oracle.spmv_csr(valA, irow, jcol, x)[k0 * kdz:k1 * kdz]
    assert np.array_equal(out[0], y) and np.array_equal(out[1], y)


# ------------------------------------------------------------------------- structured A-V form
@pytest.mark.parametrize("name", ["g2_conducting_hole_16x15x14", "g2v_conducting_moving_16x15x14",
                                  "g3_moving_coil_18x16x12", "g2i_itmax_exit_16x15x14"])
def test_structured_av_form(E, oracle, name, plane_pitch, sav_tiles):
    """ec3d_assemble's default storage for the A-V system: U embedded in the grid, every coupling a
    class-coded stencil slot, no tail.  The operator is the reference's (exported CSR and SpMV bit-identical)
    and the solve is bit-identical to the oracle's twin run on the system in device numbering -- on linear tiles,
    on runtime-shaped 2-D tiles and with the fused three-launch iteration on them."""
    if sav_tiles != "linear" and plane_pitch != "pitched":
        pytest.skip("2-D tiles need the z-marching (pitched) layout")
    g = load_golden(name)
    n = len(g["irow"]) - 1
    tol, itmax = float(g["tol"]), int(g["itmax"])
    x = np.random.Generator(np.random.PCG64(31)).standard_normal(n)
    with E.EC3DSolver() as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        mi = s.info
        assert mi.n == n and mi.tail_rows == 0 and mi.dict_classes > 27
        rm = s.row_map()
        sdz, sdy, sdx = g["geoPHYS"].shape
        nC_dev = g["geoPHYS"].size if plane_pitch == "auto" else sdz * (-(-sdx * sdy // 512) * 512)
        assert rm.max() < 4 * nC_dev and np.all(np.diff(rm) > 0)
        assert (s.geometry(1).zm_tpp > 0) == (plane_pitch == "pitched")
        assert (s.geometry(1).patch_x > 0) == (sav_tiles != "linear")
        assert s.fusion() == ((1, 1) if sav_tiles == "patch-fused" else (0, 0))
        va, ir, jc = s.export_csr()
        assert np.array_equal(ir, g["irow"]) and np.array_equal(jc, g["jcol"]) and np.array_equal(va, g["valA"])
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))
        for k in range(len(g["iters"])):
            xs, it, hist = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax, hist_cap=64)
            vd, ird, jcd, rmap, bd, xd = oracle.device_system(s, g["valA"], g["irow"], g["jcol"], g[f"b{k}"],
                                                              g[f"xin{k}"])
            xo, ito, hs, hr = oracle.bicgstab_wr_gpuorder(oracle.geoms_of(s), vd, ird, jcd, bd, xd, tol, itmax,
                                                          hist_cap=64)
            assert it == ito and np.array_equal(xs, xo[rmap])
            assert np.all(np.delete(xo, rmap) == 0.0)          # inactive U slots never move
            kk = min(it, 64)
            assert np.array_equal(hist[:kk, 0], hs[:kk])
            xr = g[f"xout{k}"]
            assert np.linalg.norm(xs - xr) <= 10 * tol * np.linalg.norm(xr)


@pytest.mark.parametrize("name", ["g2_conducting_hole_16x15x14", "g3_moving_coil_18x16x12", "g1_nonconducting_8x7x6"])
def test_structured_form_recognised_in_csr(E, oracle, name, plane_pitch, sav_tiles):
    """The drop-in route: the reference's CSR goes in, the structure is recognised entry by entry
    (ec3d_sav_csr.cpp) and the handle ends up in the same class-coded form ec3d_assemble builds natively --
    same row map, same operator bit for bit, same solve."""
    if sav_tiles != "linear" and plane_pitch != "pitched":
        pytest.skip("2-D tiles need the z-marching (pitched) layout")
    g = load_golden(name)
    n = len(g["irow"]) - 1
    tol, itmax = float(g["tol"]), int(g["itmax"])
    x = np.random.Generator(np.random.PCG64(32)).standard_normal(n)
    with E.EC3DSolver() as s, E.EC3DSolver() as nat:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        nat.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        mi = s.info
        assert mi.n == n and mi.tail_rows == 0 and mi.dict_classes > 0 and mi.nnz == len(g["valA"])
        assert np.array_equal(s.row_map(), nat.row_map())
        va, ir, jc = s.export_csr()
        assert np.array_equal(ir, g["irow"]) and np.array_equal(jc, g["jcol"]) and np.array_equal(va, g["valA"])
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))
        xs, it, hist = s.solve(g["b0"], g["xin0"], tol, itmax, hist_cap=64)
        xo, ito, hs, hr = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g["b0"], g["xin0"], tol, itmax,
                                            hist_cap=64)
        assert it == ito and np.array_equal(xs, xo) and np.array_equal(hist[:min(it, 64), 0], hs[:min(it, 64)])
        xn, itn, _ = nat.solve(g["b0"], g["xin0"], tol, itmax)
        assert itn == it and np.array_equal(xn, xs)


def synthetic_av(sdx, sdy, sdz, block, hole=None):
    """A small A-V system on any grid: one conducting block (i0, i1, j0, j1, k0, k1; 0-based, end exclusive) with an
    optional hole through it, air elsewhere; arrays as ec3d_assemble takes them (src/m_vxc2data.f90:43-52)."""
    mu0 = 0.12566370964050292e-05
    geo = np.full((sdz, sdy, sdx), 2, np.int8)
    i0, i1, j0, j1, k0, k1 = block
    geo[k0:k1, j0:j1, i0:i1] = 1
    if hole:
        a0, a1, b0, b1 = hole
        geo[k0:k1, b0:b1, a0:a1] = 2
    flat = geo.reshape(-1)
    geoC = np.zeros(flat.size, np.int32)
    idx = np.flatnonzero(flat == 1)
    geoC[idx] = 3 * flat.size + 1 + np.arange(idx.size)
    valPHYS = np.zeros((2, 5))
    valPHYS[:, 0] = 1.0
    valPHYS[0, 1] = mu0 * 35.26e6
    valPHYS[0, 2:5] = (0.3, -0.2, 0.1)          # a moving conductor: the advection terms of src/EC3D.f90:657-662
    return geo, geoC.reshape(geo.shape), valPHYS, np.array([[-0.95, -0.9], [-0.85, -0.8], [-0.75, -0.7]]), \
        np.array([0.004, 0.005, 0.003]), 1e-3


@pytest.mark.parametrize("fuse", ["0", "2"])
@pytest.mark.parametrize("dims,px,block,hole", [
    ((64, 22, 12), "32", (9, 52, 4, 19, 3, 9), (28, 36, 9, 14)),    # 2 patch columns of 32 x 16, ragged second patch row
    ((96, 21, 10), "48", (5, 90, 3, 18, 2, 8), (40, 60, 8, 13)),    # 48 x 10 (480-cell patches: idle threads), ragged rows
    ((102, 23, 9), None, (7, 95, 4, 20, 2, 7), (50, 56, 10, 14)),   # the picker's own shape for a 102-wide grid: 102 x 5
])
def test_structured_form_on_runtime_shaped_tiles(E, oracle, dims, px, block, hole, fuse, monkeypatch):
    """sav_patch_step on shapes the captured fixtures cannot reach: several patch columns, patches of fewer than 512
    cells (threads beyond the patch idle), a ragged last patch row (rows beyond the grid idle), a conductor -- with a
    hole, so every one-sided A-U stencil and Neumann-mirrored U row occurs -- that crosses patch boundaries in x and y.
    A*x == the oracle's CSR row sums of the matrix the oracle's gen_sparse_matrix builds (src/EC3D.f90:465-1049), the
    solve == the twin, bit for bit, with and without the fused launches; and the linear tiles give the same A*x."""
    sdx, sdy, sdz = dims
    geo, geoC, valPHYS, BND, delta, dt = synthetic_av(sdx, sdy, sdz, block, hole)
    m = oracle.gen_sparse_matrix(geo, geoC, valPHYS, BND, delta, dt)
    n = m["n"]
    rng = np.random.Generator(np.random.PCG64(606))
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    monkeypatch.setenv("EC3D_PITCH", "2")
    monkeypatch.setenv("EC3D_FUSE23", fuse)
    monkeypatch.setenv("EC3D_FUSE51", fuse)
    ys = {}
    for tiles in ("2", "0"):
        monkeypatch.setenv("EC3D_SAV_PATCH", tiles)
        if px and tiles == "2":
            monkeypatch.setenv("EC3D_SAV_PATCH_PX", px)
        else:
            monkeypatch.delenv("EC3D_SAV_PATCH_PX", raising=False)
        with E.EC3DSolver() as s:
            s.assemble(geo, geoC, valPHYS, BND, delta, dt)
            g1 = s.geometry(1)
            if tiles == "2":
                want_px = int(px) if px else 102
                assert (g1.patch_x, g1.patch_y, g1.patch_sdx, g1.patch_sdy) == (want_px, 512 // want_px, sdx, sdy)
                assert g1.zm_tpp == (sdx // want_px) * -(-sdy // (512 // want_px))
                assert s.fusion() == ((1, 1) if fuse == "2" else (0, 0))
            else:
                assert g1.patch_x == 0 and g1.zm_tpp > 0
            va, ir, jc = s.export_csr()
            assert np.array_equal(ir, m["irow"]) and np.array_equal(jc, m["jcol"]) and np.array_equal(va, m["valA"])
            ys[tiles] = s.spmv(x)
            assert np.array_equal(ys[tiles], oracle.spmv_csr(m["valA"], m["irow"], m["jcol"], x))
            # unpreconditioned BiCGSTAB stagnates on these little systems with a random right-hand side (the reference's
            # own solver does: itmax exit); parity does not need convergence -- 150 iterations through the itmax exit
            # (src/solvers.f90:25-29), every ||S||, ||R|| of the history and x against the twin
            xs, it, hist = s.solve(b, np.zeros(n), 1e-12, 149, hist_cap=150)
            xo, ito, hs, hr = oracle.twin_solve(s, m["valA"], m["irow"], m["jcol"], b, np.zeros(n), 1e-12, 149, hist_cap=150)
            assert it == ito == 150 and np.array_equal(xs, xo)
            assert np.array_equal(hist[:150, 0], hs[:150]) and np.array_equal(hist[:150, 1], hr[:150])
            assert s.restart_count() == oracle.last_restart_count()
            # a warm start on the same handle: the alternating P / AP buffers of the fused iteration start clean
            xw, itw, _ = s.solve(b, 0.5 * xs, 1e-12, 5)
            xwo, itwo, _, _ = oracle.twin_solve(s, m["valA"], m["irow"], m["jcol"], b, 0.5 * xs, 1e-12, 5)
            assert itw == itwo == 6 and np.array_equal(xw, xwo)
    assert np.array_equal(ys["2"], ys["0"])


def test_structured_form_needs_the_reference_row_order(E, oracle):
    """A row whose stored order is not the order the kernels add the slots in (here: two entries of one U
    row swapped) must not be taken into the structured form: bands + tail keep the stored order."""
    g = load_golden("g2_conducting_hole_16x15x14")
    valA, irow, jcol = g["valA"].copy(), g["irow"], g["jcol"].copy()
    r = len(irow) - 2  # last U row
    p = irow[r] - 1
    valA[[p, p + 1]] = valA[[p + 1, p]]
    jcol[[p, p + 1]] = jcol[[p + 1, p]]
    x = np.random.Generator(np.random.PCG64(33)).standard_normal(len(irow) - 1)
    with E.EC3DSolver() as s:
        s.set_matrix_csr(valA, irow, jcol)
        assert s.info.tail_rows > 0 and np.array_equal(s.row_map(), np.arange(s.n))
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))


def test_structured_form_falls_back(E, oracle):
    """Geometries the structured form does not cover (here: dictionary format switched off) use bands + tail."""
    g = load_golden("g2_conducting_hole_16x15x14")
    with E.EC3DSolver(dictionary=False) as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        assert s.info.tail_rows > 0
        assert np.array_equal(s.row_map(), np.arange(s.n))
