"""GPU parity tests: the HIP path (through the C ABI of libec3d_hip.so) against the oracle and the
fixtures captured from the unmodified reference.  Run on the MI355X box with ``-m gpu``.

Bars (written here, as DESIGN.md §6 states them):
  * SpMV, assembly: bit-identical to the oracle / the reference's CSR.
  * Solve vs the oracle's "GPU order" twin (same algorithm, dot products summed in the kernels'
    order): bit-identical x, iter and residual history.
  * Solve vs the reference itself: ||x - x_ref|| <= 10*tol*||x_ref||; residual history within 1e-10
    relative over the first 15 iterations; iteration counts reported side by side.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CAPTURED = ["g1_nonconducting_8x7x6", "g2_conducting_hole_16x15x14",
            "g2v_conducting_moving_16x15x14", "g3_moving_coil_18x16x12"]


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()  # raises if the HIP extension is missing: no fallback
    return E




# ------------------------------------------------------------------------------------ SpMV
@pytest.mark.parametrize("name", CAPTURED)
def test_spmv_bitwise_csr_route(E, oracle, name):
    """src/solvers.f90:54-61: DIA+tail SpMV equals the CSR row sums bit for bit."""
    g = load_golden(name)
    rng = np.random.Generator(np.random.PCG64(11))
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        for _ in range(3):
            x = rng.standard_normal(len(g["irow"]) - 1)
            assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))
        mi = s.info
        assert mi.nbands == 7 or name.startswith("g1")


@pytest.mark.parametrize("nblk", [1, 3, 8, 16])
def test_spmv_any_workgroup_count(E, oracle, nblk):
    g = load_golden("g2_conducting_hole_16x15x14")
    x = np.random.Generator(np.random.PCG64(5)).standard_normal(len(g["irow"]) - 1)
    with E.EC3DSolver(nblk=nblk) as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        assert s.geometry().nblk == nblk
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))


def test_spmv_general_csr_without_bands(E, oracle):
    """A matrix with no dominant diagonal structure goes entirely to the sliced-ELL tail."""
    rng = np.random.Generator(np.random.PCG64(3))
    n = 1500
    rows = []
    for r in range(n):
        k = int(rng.integers(0, 9))
        cols = np.sort(rng.choice(n, k, replace=False)) + 1
        rows.append(cols)
    irow = np.concatenate([[1], 1 + np.cumsum([len(c) for c in rows])]).astype(np.int32)
    jcol = np.concatenate(rows).astype(np.int32)
    valA = rng.standard_normal(len(jcol))
    x = rng.standard_normal(n)
    with E.EC3DSolver() as s:
        s.set_matrix_csr(valA, irow, jcol)
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))


# -------------------------------------------------------------------------------- assembly
@pytest.mark.parametrize("name", CAPTURED + ["g2i_itmax_exit_16x15x14"])
def test_device_assembly_equals_reference_csr(E, oracle, name):
    """src/EC3D.f90:465-1049 on the device -> exported CSR identical to the captured one."""
    g = load_golden(name)
    with E.EC3DSolver() as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        valA, irow, jcol = s.export_csr()
        assert np.array_equal(irow, g["irow"])
        assert np.array_equal(jcol, g["jcol"])
        assert np.array_equal(valA, g["valA"])
        assert s.info.nnz == len(g["jcol"])
        m = oracle.gen_sparse_matrix(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"],
                                     float(g["dt"]))
        for a, b in zip(s.cel_bnd(), m["cel_bnd"]):
            assert np.array_equal(a, b)
        # and the assembled operator acts like the reference CSR
        x = np.random.Generator(np.random.PCG64(1)).standard_normal(len(irow) - 1)
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))


@pytest.mark.parametrize("dims", [(8, 7, 6), (17, 9, 5), (33, 32, 31)])
def test_poisson_assembly_equals_oracle(E, oracle, dims):
    sdx, sdy, sdz = dims
    delta = (0.004, 0.005, 0.003)
    bnd = np.array([[-0.95, -0.9], [-0.8, -0.7], [-0.6, -0.5]])
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz, delta, np.ascontiguousarray(bnd.T).reshape(-1))
    with E.EC3DSolver() as s:
        s.assemble_poisson(sdx, sdy, sdz, delta, bnd)
        va, ir, jc = s.export_csr()
        assert np.array_equal(ir, irow) and np.array_equal(jc, jcol) and np.array_equal(va, valA)
        assert s.info.nnz == len(jcol)


def test_assembly_rejects_what_the_reference_cannot_index(E):
    g = load_golden("g2_conducting_hole_16x15x14")
    geoC = g["geoPHYS_C"].copy()
    geo = g["geoPHYS"].copy()
    geoC[0, 5, 5] = geoC.max() + 1  # conductor cell on the box boundary
    geo[0, 5, 5] = 1
    with E.EC3DSolver() as s:
        with pytest.raises(E.EC3DError):
            s.assemble(geo, geoC, g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))


# ----------------------------------------------------------------------------------- solve
# Cache policies of the kernels' streams (choose_sweep in csrc/ec3d_context.hip).  The library picks one by size --
# plain caching below 4.5 Mi rows, nontemporal streams with the next kernel's operand kept cacheable (AP, S, R = 11;
# without S = 9) up to 32 Mi rows, everything nontemporal above -- so on the small captured systems only the first
# would ever run.  EC3D_NT / EC3D_KEEP force each template instance (NT = true loads / stores, store2k's keep bits)
# onto them: every instance that ships by default meets the bitwise contract here, not only at sizes no twin can
# follow.  43 = 11 + P kept as well (a policy the sweeps of round 3 tried; it must stay correct while it is selectable).
POLICIES = [("0", None), ("1", "0"), ("1", "9"), ("1", "11"), ("1", "43")]


def set_policy(monkeypatch, policy):
    nt, keep = policy
    monkeypatch.setenv("EC3D_NT", nt)
    if keep is None:
        monkeypatch.delenv("EC3D_KEEP", raising=False)
    else:
        monkeypatch.setenv("EC3D_KEEP", keep)


@pytest.mark.parametrize("policy", POLICIES, ids=lambda p: f"nt{p[0]}-keep{p[1]}")
@pytest.mark.parametrize("name", CAPTURED)
def test_solve_bitwise_vs_gpu_order_oracle(E, oracle, name, policy, plane_pitch, monkeypatch):
    """Every captured call (warm starts included): x, iter and the whole residual history are
    bit-identical to the oracle run with the kernels' summation order -- under every cache policy the
    library can select, on the default and on the pitched (z-marching) structured form."""
    set_policy(monkeypatch, policy)
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        for k in range(len(g["iters"])):
            x, it, hist = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax, hist_cap=400)
            xo, ito, hs, hr = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g[f"b{k}"],
                                                          g[f"xin{k}"], tol, itmax, hist_cap=400)
            assert it == ito
            assert np.array_equal(x, xo)
            assert np.array_equal(hist[:it, 0], hs[:it])
            if it > 1:
                assert np.array_equal(hist[:it - 1, 1], hr[:it - 1])


@pytest.mark.parametrize("policy", POLICIES[1:], ids=lambda p: f"nt{p[0]}-keep{p[1]}")
@pytest.mark.parametrize("name", ["g2_conducting_hole_16x15x14", "g3_moving_coil_18x16x12"])
def test_structured_two_dimensional_tiles_every_policy(E, oracle, name, policy, sav_tiles, monkeypatch):
    """The structured A-V kernels on runtime-shaped 2-D tiles (sav_patch_step), unfused and with K2 inside K3 / K5
    inside the next K1, under the nontemporal policies their sizes run with by default: every captured call bit-identical
    to the twin."""
    if sav_tiles == "linear":
        pytest.skip("covered by test_solve_bitwise_vs_gpu_order_oracle")
    set_policy(monkeypatch, policy)
    monkeypatch.setenv("EC3D_PITCH", "2")
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        assert s.geometry(1).patch_x > 0 and s.fusion() == ((1, 1) if sav_tiles == "patch-fused" else (0, 0))
        for k in range(len(g["iters"])):
            x, it, hist = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax, hist_cap=400)
            xo, ito, hs, hr = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g[f"b{k}"], g[f"xin{k}"], tol, itmax,
                                                hist_cap=400)
            assert it == ito and np.array_equal(x, xo)
            assert np.array_equal(hist[:it, 0], hs[:it])
            if it > 1:
                assert np.array_equal(hist[:it - 1, 1], hr[:it - 1])


@pytest.mark.parametrize("name", CAPTURED)
def test_solve_vs_reference_fixture(E, name):
    """Against the unmodified reference's outputs: fields within 10*tol, iterations side by side."""
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        for k, it_ref in enumerate(g["iters"]):
            x, it, _ = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            xr = g[f"xout{k}"]
            rel = np.linalg.norm(x - xr) / np.linalg.norm(xr)
            print(f"{name} step {k}: iter gpu {it} / reference {int(it_ref)}, rel diff {rel:.2e}")
            assert rel <= 10 * tol
            if name.startswith(("g2", "g3")):
                # tol 5e-3 / 1e-3, 14-79 iterations: too few for the re-associated dot products to move a
                # decision -- the count IS the reference's
                assert it == int(it_ref)
            else:
                # g1 runs to tol 1e-6 (27-33 iterations with a residual plateau before the exit): the last
                # step lands 2 iterations later than the reference today; bounded, as on the tol-1e-8 cubes
                assert abs(it - int(it_ref)) <= max(3, int(0.15 * int(it_ref)))


@pytest.mark.parametrize("N", [16, 32, 64])
def test_cube_residual_history_vs_reference(E, oracle, N):
    """G5 cubes (tol 1e-8): ||R_k|| of the unmodified solver for k <= 24; the GPU history must agree
    to 1e-10 relative over the first 15 iterations (BASELINE.md §2c explains why not forever)."""
    g = load_golden(f"g5_cube{N}")
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x, it, hist = s.solve(oracle.bar_rhs(N), np.zeros(N ** 3), 1e-8, 100000, hist_cap=24)
    ref = g["rnorm_first"]
    rel = np.abs(hist[:24, 1] - ref) / ref
    window = int(np.argmax(rel > 1e-10)) if np.any(rel > 1e-10) else 24
    print(f"cube {N}: iter gpu {it} / reference {int(g['iter'])}; history agrees to 1e-10 for {window} iterations"
          f" (max rel over first 15: {rel[:15].max():.2e})")
    assert rel[:15].max() <= 1e-10
    assert np.linalg.norm(x) == pytest.approx(float(g["xnorm"]), rel=1e-6)
    # the iteration COUNT at tol 1e-8 is not a stable quantity of this algorithm (unpreconditioned
    # BiCGSTAB, chaotic under re-association): the reference's own -O3 -ffast-math build moves it
    # 270 -> 297 and 603 -> 699 and our two launch geometries give 603 -> 640 / 815
    # at 64^3.  Reported above; only sanity-bounded here.  Convergence itself is asserted via ||x||.
    assert 0.5 * int(g["iter"]) <= it <= 2 * int(g["iter"])


def test_dropin_symbol_with_warm_starts(E):
    """sprsbcgstabwr_ itself (F77 ABI), called the way src/EC3D.f90:408 does, step after step."""
    g = load_golden("g3_moving_coil_18x16x12")
    tol, itmax = float(g["tol"]), int(g["itmax"])
    n = len(g["irow"]) - 1
    x = g["xin0"].copy()
    for k, it_ref in enumerate(g["iters"]):
        assert np.array_equal(x, g[f"xin{k}"]) or k > 0
        x = g[f"xin{k}"].copy()  # the reference mutates Uaf between calls (src/EC3D.f90:428-432)
        it = E.sprsBCGstabWR(g["valA"], g["irow"], g["jcol"], n, g[f"b{k}"], x, tol, itmax)
        rel = np.linalg.norm(x - g[f"xout{k}"]) / np.linalg.norm(g[f"xout{k}"])
        assert rel <= 10 * tol and abs(it - int(it_ref)) <= max(3, int(0.15 * int(it_ref)))
    E.load_library().ec3d_invalidate()


def test_dropin_symbol_notices_a_matrix_rebuilt_in_place(E, oracle):
    """the drop-in caches the device matrix across calls (the reference assembles once,
    src/EC3D.f90:115).  A caller that changes ONE coefficient in place, at the same addresses and not among the few
    thousand sampled entries, must still get the answer of the matrix it passes: every entry is hashed (on host
    threads, while the GPU already solves on the cached matrix) and a mismatch discards that solve."""
    N = 24
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    n, nnz = N ** 3, len(valA)
    assert nnz // 4096 > 1                       # so that most entries are NOT in the quick sample
    b = oracle.bar_rhs(N)
    x1 = np.zeros(n)
    it1 = E.sprsBCGstabWR(valA, irow, jcol, n, b, x1, 1e-8, 5000)
    p = (irow[n // 2] - 1) + 3                   # an entry of a middle row ...
    step = nnz // 4096
    while p % step == 0:                         # ... that the sample does not look at
        p += 1
    valA[p] *= 1.5                               # in place: same array, same address
    x2 = np.zeros(n)
    it2 = E.sprsBCGstabWR(valA, irow, jcol, n, b, x2, 1e-8, 5000)
    with E.EC3DSolver() as s:                    # what a fresh conversion of the changed matrix gives
        s.set_matrix_csr(valA, irow, jcol)
        xr, itr, _ = s.solve(b, np.zeros(n), 1e-8, 5000)
    assert it2 == itr and np.array_equal(x2, xr)
    assert not np.array_equal(x2, x1)
    E.load_library().ec3d_invalidate()


# ------------------------------------------------------------------------------ edge cases
def test_zero_rhs_returns_immediately(E):
    """src/solvers.f90:23: ||b|| = 0 -> iter = 0, x untouched."""
    g = load_golden("g1_nonconducting_8x7x6")
    n = len(g["irow"]) - 1
    x0 = np.random.Generator(np.random.PCG64(9)).standard_normal(n)
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        x, it, _ = s.solve(np.zeros(n), x0, 1e-6, 100)
        assert it == 0 and np.array_equal(x, x0)


def test_itmax_exit_matches_reference(E, oracle, capfd):
    """src/solvers.f90:25-29: itmax = 25 -> 26 iterations, ||R|| printed, x as the reference's."""
    g = load_golden("g2i_itmax_exit_16x15x14")
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        x, it, hist = s.solve(g["b0"], g["xin0"], float(g["tol"]), int(g["itmax"]), hist_cap=26)
        assert it == 26 == int(g["iters"][0])
        xo, ito, _, _ = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g["b0"], g["xin0"],
                                                    float(g["tol"]), int(g["itmax"]))
        assert np.array_equal(x, xo)
        assert np.linalg.norm(x - g["xout0"]) <= 1e-6 * np.linalg.norm(g["xout0"])
    # the printed ||R|| (src/solvers.f90:27): the last iteration's norm, written the way the reference's toolchain
    # writes a REAL(8) in list-directed output (leading blank, shortest digits; tests/test_oracle_golden.py pins the format)
    import ctypes as C
    line = capfd.readouterr().out.rstrip("\n")
    buf = C.create_string_buffer(64)
    E.load_library().ec3d_format_real8(float(hist[25, 1]), buf)
    assert line == buf.value.decode() and line.startswith(" ") and float(line.replace("E", "e")) == hist[25, 1]


def test_loose_tolerance_takes_the_s_exit(E, oracle):
    """||S||/||b|| < tol in the first iteration: X += alpha*P and exit (src/solvers.f90:34-38)."""
    g = load_golden("g1_nonconducting_8x7x6")
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        x, it, hist = s.solve(g["b0"], g["xin0"], 0.9, 100, hist_cap=4)
        xo, ito, hs, _ = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g["b0"], g["xin0"],
                                                     0.9, 100, hist_cap=4)
        assert it == ito == 1 and np.array_equal(x, xo) and hist[0, 0] == hs[0]


def test_restart_rule_is_exercised(E, oracle):
    """The restart |R.R0|/||b|| < tol (src/solvers.f90:47-49) fires in the captured runs; make sure
    the device path takes it identically (bitwise parity across a restart)."""
    g = load_golden("g1_nonconducting_8x7x6")
    tol = 3e-2
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        x, it, _ = s.solve(g["b0"], g["xin0"], tol, 1000)
        xo, ito, _, _ = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g["b0"], g["xin0"],
                                                    tol, 1000)
        assert it == ito and np.array_equal(x, xo)
        assert s.restart_count() == oracle.last_restart_count() > 0


# ----------------------------------------------------------------- full-size, size-independent
def test_full_size_256_known_answer_and_linearity(E):
    """BASELINE config 2 size (256^3, n = 16 777 216): b = A x*, x* ~ U(-1,1) PCG64(12345)
    (SURVEY §8d).  Checks SpMV linearity and that the solve drives the true residual below tol."""
    N = 256
    n = N ** 3
    rng = np.random.Generator(np.random.PCG64(12345))
    xs = rng.uniform(-1.0, 1.0, n)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        assert s.info.nnz == 7 * n - 6 * N * N
        b = s.spmv(xs)
        y = rng.uniform(-1.0, 1.0, n)
        lin = s.spmv(2.0 * xs - 0.5 * y) - (2.0 * b - 0.5 * s.spmv(y))
        assert np.abs(lin).max() <= 1e-9 * np.abs(b).max()
        tol = 1e-6
        x, it, hist = s.solve(b, np.zeros(n), tol, 20000, hist_cap=8)
        res = np.linalg.norm(b - s.spmv(x)) / np.linalg.norm(b)
        print(f"256^3 known answer: iter {it}, true residual {res:.2e}, "
              f"error {np.linalg.norm(x - xs) / np.linalg.norm(xs):.2e}")
        assert res <= 2 * tol


def test_full_size_256_spmv_bitwise_vs_oracle(E, oracle):
    N = 256
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    x = np.random.Generator(np.random.PCG64(77)).standard_normal(N ** 3)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))


@pytest.mark.parametrize("policy", POLICIES, ids=lambda p: f"nt{p[0]}-keep{p[1]}")
@pytest.mark.parametrize("fuse", ["0", "2"])
@pytest.mark.parametrize("grid", [(128, 16, 12), (256, 8, 9), (128, 12, 10)])
def test_two_dimensional_tiles_bitwise(E, oracle, grid, fuse, policy, monkeypatch):
    """The 2-D tiles of the z-marching dictionary kernels (patch_pair in csrc/ec3d_kernels.hip: a workgroup owns a
    128 x 4 patch of the xy plane, the +-sdx neighbours travel through LDS): A*x equals the oracle's CSR SpMV bit for
    bit (src/solvers.f90:54-61), the solve equals the oracle's GPU-order twin bit for bit -- x, iter and the whole
    residual history -- with the thread -> cell assignment the library reports (ec3d_geom::patch_x), and the same
    system on 512-consecutive-cell tiles (EC3D_PATCH=0) gives the same A*x bit for bit and the same solution to
    rounding growth.  Grids: two patch rows of 4 / one patch column; a 256-wide grid (two patch columns); 12 rows.
    fuse = 2: K2 inside K3 (k23_s_spmv_dots, the default from 20 Mi rows -- z-slabs 32 Mi --, forced here): S.S is then summed in the SpMV
    kernels' order, which the library reports as geometry 2 -- the twin must still match bit for bit."""
    monkeypatch.setenv("EC3D_FUSE23", fuse)
    monkeypatch.setenv("EC3D_FUSE51", fuse)      # and K5 inside the next K1 (k51_p_spmv_dot), P / AP alternating buffers
    if policy != POLICIES[0] and grid != (256, 8, 9):
        pytest.skip("cache policies: one grid is enough (the policy does not depend on the grid)")
    set_policy(monkeypatch, policy)
    sdx, sdy, sdz = grid
    n = sdx * sdy * sdz
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(2026))
    x0 = rng.standard_normal(n)
    b = rng.standard_normal(n)
    tol = 1e-9
    res = {}
    for patch in ("1", "0"):
        monkeypatch.setenv("EC3D_PATCH", patch)
        with E.EC3DSolver() as s:
            s.assemble_poisson(sdx, sdy, sdz)
            g = s.geometry(1)
            assert g.zm_tpp == sdx * sdy // 512
            assert (g.patch_x, g.patch_y, g.patch_sdx) == ((128, 4, sdx) if patch == "1" else (0, 0, 0))
            assert (s.geometry(2).nblk == g.nblk and s.geometry(2).patch_x == 128) == (patch == "1" and fuse == "2")
            assert np.array_equal(s.spmv(x0), oracle.spmv_csr(valA, irow, jcol, x0))
            x, it, hist = s.solve(b, np.zeros(n), tol, 5000, hist_cap=64)
            xo, ito, hs, hr = oracle.twin_solve(s, valA, irow, jcol, b, np.zeros(n), tol, 5000, hist_cap=64)
            m = min(it, 64)
            assert it == ito and np.array_equal(x, xo)
            assert np.array_equal(hist[:m, 0], hs[:m]) and np.array_equal(hist[:m - 1, 1], hr[:m - 1])
            # the restart R0 = R, P = R (src/solvers.f90:47-49) fired -- with fuse = 2 inside k51_p_spmv_dot, which then
            # rewrites R0 in the SpMV kernel -- as often as in the twin (counted by the device / by the oracle)
            assert s.restart_count() == oracle.last_restart_count() > 0
            res[patch] = (x, it)
            # the same handle again: a warm start that runs into itmax (src/solvers.f90:25-29: 8 iterations), then a
            # solve from the converged x (exits at once) -- the alternating P / AP buffers of the fused iteration
            # must start every solve in the same state
            xw, itw, hw = s.solve(b, 0.5 * x, tol, 7, hist_cap=16)
            xwo, itwo, hws, hwr = oracle.twin_solve(s, valA, irow, jcol, b, 0.5 * x, tol, 7, hist_cap=16)
            assert itw == itwo == 8 and np.array_equal(xw, xwo) and np.array_equal(hw[:8, 0], hws[:8])
            xc, itc, _ = s.solve(b, x, tol, 5000)
            xco, itco, _, _ = oracle.twin_solve(s, valA, irow, jcol, b, x, tol, 5000)
            assert itc == itco and np.array_equal(xc, xco)
    assert np.linalg.norm(res["1"][0] - res["0"][0]) <= 1e-6 * np.linalg.norm(res["0"][0])


@pytest.mark.parametrize("grid", [(256, 8, 9), (128, 16, 12), (128, 12, 10)], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("fuse", ["2", "2s", "0"], ids=["three-launches", "three-launches-K4-as-SpMV", "five-launches"])
@pytest.mark.parametrize("depth", [1, 2, 3, 4])
def test_deferred_x_update_bitwise(E, oracle, depth, fuse, grid, monkeypatch):
    monkeypatch.setenv("EC3D_XASYNC", "0")      # the K4 of a group's last iteration applies the group itself
    deferred_x_case(E, oracle, depth, fuse, grid, monkeypatch, False)


@pytest.mark.parametrize("grid", [(256, 8, 9), (128, 12, 10)], ids=lambda g: "x".join(map(str, g)))
@pytest.mark.parametrize("depth", [2, 3, 4])
def test_deferred_x_update_as_launches_of_the_iterations_stream_bitwise(E, oracle, depth, grid, monkeypatch):
    """The default of an undivided handle on the three-launch iteration with K4 in SpMV form (from 20 Mi rows; forced on a
    small grid here, no EC3D_XASYNC in the environment): no K4 touches X, every group of `depth` updates is applied by a
    launch of its own (k_x_group) on the iteration's OWN stream behind the K4 of the group's last iteration, P and S in
    rings of ONE group -- every exit at every position of a group, the itmax exit, bench-style calls: the twin's bits."""
    monkeypatch.delenv("EC3D_XASYNC", raising=False)
    deferred_x_case(E, oracle, depth, "2s", grid, monkeypatch, 2)


@pytest.mark.parametrize("fuse", ["2", "2s", "0"], ids=["three-launches", "three-launches-K4-as-SpMV", "five-launches"])
@pytest.mark.parametrize("depth", [2, 3, 4])
def test_deferred_x_update_on_a_second_stream_bitwise(E, oracle, depth, fuse, monkeypatch):
    """The same cases with every group of `depth` updates applied by a launch of its own (k_x_group) on a second stream,
    beside the iterations that follow (EC3D_XASYNC=2 forces it on a plain handle; meant for z-slabs): no K4 touches X, P and
    S wait in rings of two groups, alpha / omega in entry it % (2 depth) of the device state; the group an exit lies in
    ends there by itself (or is added by the host when it had not been enqueued yet), the last group of a call is cut
    short and joined.  Eight workgroups, so the launch's stride differs from the vector kernels' grid."""
    monkeypatch.setenv("EC3D_XASYNC", "2")
    monkeypatch.setenv("EC3D_XASYNC_WGS", "8")
    deferred_x_case(E, oracle, depth, fuse, (256, 8, 9), monkeypatch, True)


def deferred_x_case(E, oracle, depth, fuse, grid, monkeypatch, second_stream):
    """X = X + alpha*P + omega*S (src/solvers.f90:41) applied every `depth`-th iteration (k4d_x_r_update: P and S of the
    pending iterations wait in rings, alpha and omega in the solver state; the default from 32 Mi rows with depth 4,
    forced here on the three-launch iteration of a small grid).  Nothing in the loop reads X, the updates are applied in
    order and each as its own two rounded additions: x must be the twin's bit for bit -- after the ||R|| exit and the
    ||S|| exit at every position in a group (the pending updates are then applied by k_x_flush before the solve
    returns, the ||S|| exit's X = X + alpha*P as a half update), after the itmax exit at every position (the last
    iteration applies what is pending), and after bench-style ec3d_iterate calls of any length."""
    # "2s": K4 as an SpMV kernel (k4s_x_r_spmv) that computes AS = A S again instead of reading what K23 no longer writes
    # -- the default from 32 Mi rows; R.R and R.R0 are then summed in the SpMV kernels' order (geometry 0 says so)
    k4s = fuse == "2s"
    fuse = fuse[0]
    if depth == 1 and not k4s:
        pytest.skip("the classic K4: every other parity test")
    monkeypatch.setenv("EC3D_K4S", "2" if k4s else "0")
    monkeypatch.setenv("EC3D_FUSE23", fuse)     # five launches: K2 writes S, K5 the new P into the next buffer of the rings
    monkeypatch.setenv("EC3D_FUSE51", fuse)
    monkeypatch.setenv("EC3D_PATCH", "1")
    monkeypatch.setenv("EC3D_XDEFER", str(depth))
    sdx, sdy, sdz = grid
    n = sdx * sdy * sdz
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(414 + sdy))
    x0 = rng.standard_normal(n)
    b = rng.standard_normal(n)
    with E.EC3DSolver() as s:
        s.assemble_poisson(sdx, sdy, sdz)
        assert s.fusion() == ((1, 1) if fuse == "2" else (0, 0)) and s.x_interval() == depth
        assert s.x_groups()[0] == (int(second_stream) if depth > 1 else 0)
        assert (s.geometry(0).nblk == s.geometry(1).nblk and s.geometry(0).patch_x == 128) == k4s
        # to convergence, with the history: the norms below give tolerances that end the solve at chosen iterations
        x, it, hist = s.solve(b, x0, 1e-10, 5000, hist_cap=64)
        xo, ito, hs, hr = oracle.twin_solve(s, valA, irow, jcol, b, x0, 1e-10, 5000, hist_cap=64)
        assert it == ito and np.array_equal(x, xo) and it > 16
        assert np.array_equal(hist[:16, 0], hs[:16]) and np.array_equal(hist[:16, 1], hr[:16])
        bnorm = float(np.linalg.norm(b))
        seen = set()
        for k in range(1, 14):
            for col in (0, 1):          # just above ||S_k|| / ||b|| and ||R_k|| / ||b||: an exit at iteration <= k
                tol = float(hist[k - 1, col]) / bnorm * (1 + 1e-9)
                xe, ite, _ = s.solve(b, x0, tol, 5000)
                xeo, iteo, hse, hre = oracle.twin_solve(s, valA, irow, jcol, b, x0, tol, 5000, hist_cap=32)
                assert ite == iteo and np.array_equal(xe, xeo), (depth, k, col)
                s_exit = hse[ite - 1] / bnorm < tol
                seen.add(((ite - 1) % depth, "S" if s_exit else "R"))
        # both exits met at every position of a group (asserted on the grid where the histories are known to do so)
        if grid == (256, 8, 9):
            assert seen == {(m, kind) for m in range(depth) for kind in "SR"}, seen
        for itmax in range(0, 2 * depth + 2):       # src/solvers.f90:25-29 after 1 .. 2 depth + 2 iterations
            xm, itm, _ = s.solve(b, x0, 1e-30, itmax)
            xmo, itmo, _, _ = oracle.twin_solve(s, valA, irow, jcol, b, x0, 1e-30, itmax)
            assert itm == itmo == itmax + 1 and np.array_equal(xm, xmo), (depth, itmax)
        # bench steps (exits and restart disabled): calls of 5 and 3 iterations leave the X of 8 iterations
        s.upload("B", b)
        s.upload("X", x0)
        s.iterate_begin()
        s.iterate(1, 5)
        s.iterate(6, 3)
        s.synchronize()
        xi = s.download("X")
        xr, itr, _ = s.solve(b, x0, 0.0, 7)          # tol = 0: no exit, no restart either -- 8 iterations
        assert itr == 8 and np.array_equal(xi, xr)


@pytest.mark.parametrize("name", CAPTURED)
def test_deferred_x_update_on_the_captured_systems(E, oracle, name, plane_pitch, sav_tiles, monkeypatch):
    """The same on the reference's own systems [Ax | Ay | Az | U] (structured form, default and pitched; five launches
    on linear and on 2-D tiles, three launches on 2-D tiles): every captured call with X applied every fourth iteration
    -- x, iter and the history are the twin's."""
    if sav_tiles != "linear" and plane_pitch != "pitched":
        pytest.skip("the 2-D tiles of the structured kernels need tile-aligned planes")
    monkeypatch.setenv("EC3D_XDEFER", "4")
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        assert s.x_interval() == 4
        for k in range(len(g["iters"])):
            x, it, hist = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax, hist_cap=400)
            xo, ito, hs, hr = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g[f"b{k}"], g[f"xin{k}"], tol, itmax,
                                                hist_cap=400)
            assert it == ito and np.array_equal(x, xo)
            assert np.array_equal(hist[:it, 0], hs[:it])


def test_large_grid_768_formats_bitwise(E, monkeypatch):
    """Well beyond the benchmark size (768^3, n = 452 984 832, 3.2e9 nonzeros > 2^31): plain DIA streams
    (25 GB) and the dictionary form give bit-identical iterates after 6 iterations -- a size-independent
    check that the large-index paths (64-bit row arithmetic, z-marching map with 2 z-segments, 3
    workgroups per CU) agree.  x of the two runs is compared on the device side via checksums."""
    monkeypatch.setenv("EC3D_PATCH", "0")     # same tiles for both formats: the dictionary's 2-D tiles sum the dots in another order
    N = 768
    n = N ** 3
    rng = np.random.Generator(np.random.PCG64(99))
    b = np.zeros(n)
    idx = rng.integers(0, n, 200000)
    b[idx] = rng.standard_normal(idx.size)
    out = []
    for dic in (False, True):
        with E.EC3DSolver(dictionary=dic) as s:
            s.assemble_poisson(N, N, N)
            assert s.info.nnz == 7 * n - 6 * N * N
            assert s.geometry(1).zm_tpp == N * N // 512
            x, it, hist = s.solve(b, np.zeros(n), 1e-30, 5, hist_cap=6)   # itmax exit after 6 iterations
            assert it == 6 and np.all(np.isfinite(hist)) and hist[5, 1] < hist[0, 1]
            out.append((x, hist.copy()))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])
