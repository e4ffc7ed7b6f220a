"""Pin the oracle (oracle/ec3d_oracle.c) against fixtures captured from the UNMODIFIED reference
(tests/golden/*.npz, written by oracle/make_goldens.py with oracle/_ref/EC3D_capture).
CPU only.  Bar: bit-identical iteration counts, solutions and CSR arrays."""
import numpy as np
import pytest

from conftest import load_golden

CAPTURED = ["g1_nonconducting_8x7x6", "g2_conducting_hole_16x15x14",
            "g2v_conducting_moving_16x15x14", "g2i_itmax_exit_16x15x14", "g3_moving_coil_18x16x12"]


@pytest.mark.parametrize("name", CAPTURED)
def test_solver_restatement_bitwise(oracle, name):
    """src/solvers.f90:3-50 vs oracle_bicgstab_wr on every captured call (warm starts included)."""
    g = load_golden(name)
    for s, it_ref in enumerate(g["iters"]):
        x, it, _, _ = oracle.bicgstab_wr(g["valA"], g["irow"], g["jcol"], g[f"b{s}"], g[f"xin{s}"],
                                         float(g["tol"]), int(g["itmax"]))
        assert it == int(it_ref)
        assert np.array_equal(x, g[f"xout{s}"])


def test_itmax_exit_runs_itmax_plus_one(oracle):
    """src/solvers.f90:25-29: the test precedes the increment, so itmax=25 gives 26 iterations."""
    g = load_golden("g2i_itmax_exit_16x15x14")
    assert int(g["itmax"]) == 25 and all(int(i) == 26 for i in g["iters"])


@pytest.mark.parametrize("name", CAPTURED)
def test_assembly_restatement_bitwise(oracle, name):
    """src/EC3D.f90:465-1049 vs oracle_gen_sparse_matrix: irow, jcol, valA identical."""
    g = load_golden(name)
    m = oracle.gen_sparse_matrix(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"],
                                 float(g["dt"]))
    assert np.array_equal(m["irow"], g["irow"])
    assert np.array_equal(m["jcol"], g["jcol"])
    assert np.array_equal(m["valA"], g["valA"])


def test_assembly_covers_every_u_branch(oracle):
    """The G2 block with a through-hole exercises corner/edge/face/interior U rows (7 or 13
    entries) and both one-sided A-U stencils (rows of 10)."""
    g = load_golden("g2_conducting_hole_16x15x14")
    hist = np.bincount(np.diff(g["irow"]), minlength=14)
    assert hist[13] > 0 and hist[10] > 0 and hist[9] > 0 and hist[7] > 0
    m = oracle.gen_sparse_matrix(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"],
                                 float(g["dt"]))
    assert all(len(c) > 0 for c in m["cel_bnd"])


def test_rhs_zeroed_at_onesided_cells(oracle):
    """src/EC3D.f90:396-402 zeroes Jaf at cel_bnd*: the captured b must be 0 exactly there."""
    g = load_golden("g2_conducting_hole_16x15x14")
    m = oracle.gen_sparse_matrix(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"],
                                 float(g["dt"]))
    for lst in m["cel_bnd"]:
        assert np.all(g["b1"][lst - 1] == 0.0)


@pytest.mark.parametrize("N", [16, 32])
def test_cube_iterations_and_norm(oracle, N):
    """G5: reference solver alone on the config-2 operator: iter and ||x||."""
    g = load_golden(f"g5_cube{N}")
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    x, it, hs, hr = oracle.bicgstab_wr(valA, irow, jcol, oracle.bar_rhs(N), np.zeros(N ** 3), 1e-8,
                                       100000, hist_cap=24)
    assert it == int(g["iter"])
    assert np.linalg.norm(x) == pytest.approx(float(g["xnorm"]), rel=1e-14)
    if N <= 32:
        assert np.array_equal(x, g["x"])
    # residual history of the unmodified solver (itmax trick, src/solvers.f90:25-28)
    assert np.allclose(hr, g["rnorm_first"], rtol=1e-13, atol=0)


def test_cube_iterates_match_reference(oracle):
    """x after exactly k iterations (k = 1..24) equals the reference's, via the itmax exit."""
    g = load_golden("g5_cube16")
    N = 16
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    b = oracle.bar_rhs(N)
    for k in (1, 2, 7, 24):
        x, it, _, _ = oracle.bicgstab_wr(valA, irow, jcol, b, np.zeros(N ** 3), 1e-300, k - 1)
        assert it == k
        assert np.array_equal(x[g["probes"]], g["xk_probe"][k - 1])
        assert np.linalg.norm(x) == pytest.approx(float(g["xk_norm"][k - 1]), rel=1e-14)


def test_known_survey_numbers():
    """SURVEY §8c [measured] values for the shipped input and cubes are what our captures show."""
    g = load_golden("g4_compare_to_Elmer")
    assert list(g["iters"]) == [173, 160, 80]
    assert int(g["n"]) == 792288 and int(g["nnz"]) == 5892072
    assert np.allclose(g["bnorm"], [50.79818, 58.99444, 65.39831], rtol=1e-6)
    assert np.allclose(g["xnorm"], [1.219872e-2, 1.223487e-2, 1.100793e-2], rtol=1e-6)
    assert list(g["rowlen_hist"][[4, 5, 6, 7, 9, 10, 13]]) == [24, 2664, 86400, 546704, 112320, 17280, 26896]
    assert int(load_golden("g5_cube32")["iter"]) == 270 and int(load_golden("g5_cube64")["iter"]) == 603


def test_norm2_and_dot_semantics(oracle):
    rng = np.random.Generator(np.random.PCG64(1))
    a = rng.standard_normal(1000) * 10.0 ** rng.integers(-3, 3, 1000)
    assert oracle.norm2(a) == pytest.approx(np.linalg.norm(a), rel=1e-14)
    assert oracle.norm2(np.zeros(5)) == 0.0
    s = 0.0
    for u, v in zip(a, a[::-1]):
        s = s + u * v
    assert oracle.dot(a, a[::-1].copy()) == s


def test_gpu_order_dot_is_a_permutation_of_the_sum(oracle):
    """The GPU-order twin only re-associates: same value to ~1e-15, and deterministic."""
    rng = np.random.Generator(np.random.PCG64(2))
    n = 5000
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    for nblk, xg in ((3, 0), (16, 2), (64, 8)):
        geom = oracle.GpuGeom(n_pad=5120, tile=512, nblk=nblk, threads=256, xcd_group=xg)
        d = oracle.dot_gpuorder(geom, a, b)
        assert d == oracle.dot_gpuorder(geom, a, b)
        assert d == pytest.approx(float(np.dot(a, b)), rel=1e-12)


def test_count_sketch_is_linear_and_estimates_distances(oracle):
    """oracle.count_sketch: what lets the full-size tests state ||x_gpu - x_ref|| / ||x_ref|| from a 32 KB fixture."""
    rng = np.random.Generator(np.random.PCG64(9))
    n = 1_500_000
    x = rng.standard_normal(n)
    y = x + 0.03 * rng.standard_normal(n) * np.linspace(0, 2, n)
    sx, sy = oracle.count_sketch(x), oracle.count_sketch(y)
    assert np.allclose(oracle.count_sketch(2.5 * x - y), 2.5 * sx - sy, rtol=0, atol=1e-9 * np.abs(sx).max())
    est = np.linalg.norm(sx - sy) / np.linalg.norm(sx)
    true = np.linalg.norm(x - y) / np.linalg.norm(x)
    assert abs(est - true) <= 0.06 * true                    # ~1 % standard deviation at 4096 buckets
    assert np.array_equal(sx, oracle.count_sketch(x))       # pure index arithmetic: reproducible anywhere


def test_full_size_fixtures_hold_what_the_gpu_tests_read():
    import os
    from conftest import GOLDEN
    for name in ("g6_ec_src_move_hole_256x256x60", "g6_LIM_384x192x128"):
        if not os.path.exists(os.path.join(GOLDEN, name + ".npz")):
            continue
        g = load_golden(name)
        k = len(g["iters"])
        assert g["xsketch"].shape == (k, 4096) and g["xprobe"].shape == (k, 200) and g["bprobe"].shape == (k, 200)
        assert int(g["n"]) == 3 * int(np.prod(g["dims"])) + (int(g["n"]) - 3 * int(np.prod(g["dims"])))
        assert all(f"vtk_field_{N}_Field_A_sketch" in g.files for N in range(1, k - 1))


def test_itmax_line_is_written_the_way_the_reference_toolchain_writes_it(oracle):
    """`print*, norm2(R)` (src/solvers.f90:27): the oracle and the library format that number as flang's list-directed
    output does -- 217 doubles against what a program compiled with amdflang printed for them
    (tests/golden/flang_list_directed.json, oracle/make_list_directed_fixture.py), character for character."""
    import ctypes as C
    import json
    import os
    from conftest import GOLDEN
    import eddy_currents_3d_amd as E
    fx = json.load(open(os.path.join(GOLDEN, "flang_list_directed.json")))
    Lo, Ll = oracle.lib(), E.load_library()
    Lo.oracle_format_list_directed.argtypes = [C.c_double, C.c_char_p]
    Lo.oracle_format_list_directed.restype = None
    for hx, want in zip(fx["values_hex"], fx["text"]):
        v = float.fromhex(hx)
        for fn in (Lo.oracle_format_list_directed, Ll.ec3d_format_real8):
            buf = C.create_string_buffer(64)
            fn(v, buf)
            assert buf.value.decode() == want, (hx, fn)


def test_itmax_line_in_gfortran_style():
    """List-directed output is compiler specific: the reference's own Makefile builds with gfortran (src/Makefile:1-28),
    which writes a REAL(8) as one blank and G25.17E3.  EC3D_PRINT_STYLE=gfortran selects this formatter for the itmax line
    (src/solvers.f90:27); no gfortran exists in this image, so the expectations are the edit descriptor's rules: 17
    significant digits, F editing with five trailing blanks for 0.1 <= |x| < 10^17, d.dddE+eee otherwise, 26 characters."""
    import ctypes as C
    import eddy_currents_3d_amd as E
    L = E.load_library()
    L.ec3d_format_real8_gfortran.argtypes = [C.c_double, C.c_char_p]
    L.ec3d_format_real8_gfortran.restype = None

    def f(v):
        buf = C.create_string_buffer(64)
        L.ec3d_format_real8_gfortran(v, buf)
        return buf.value.decode()
    assert f(0.5813987794206226) == "  0.58139877942062257     "
    assert f(1.0e-5) == "   1.0000000000000001E-005"
    assert f(1.0) == "   1.0000000000000000     "
    assert f(-16.27049629976871) == "  -16.270496299768709     "
    assert f(0.0) == "   0.0000000000000000     "
    assert f(1.0e17).strip() == "1.0000000000000000E+017"
    assert f(12345678901234567.0) == "   12345678901234568.     "       # 17 digits in front of the point: the point stays
    assert f(1.0e16) == "   10000000000000000.     "
    for v in (3.14159, 2.5e10, 7e-300, 123456789.125, 0.1, 0.099999999):
        s = f(v)
        assert len(s) == 26 and float(s) == v


class _FakeSlab:
    """Stands in for an EC3DSolver view of a slab (what oracle.geoms_of asks of it): one launch geometry for every kernel."""

    def __init__(self, oracle, n_rows, nblk, xg=0):
        self.O, self.n_rows, self.nblk, self.xg = oracle, n_rows, nblk, xg

    def ulist(self):
        import numpy as np
        return np.zeros(0, np.int32)

    def geometry(self, which):
        n_pad = (self.n_rows + 511) // 512 * 512
        return self.O.GpuGeom(n_pad=n_pad, tile=512, nblk=self.nblk, threads=256, xcd_group=self.xg)


def test_multi_rank_twin_on_one_rank_is_the_single_rank_twin(oracle):
    """oracle.twin_solve_slabs -- the whole system, every dot product summed rank by rank in the ranks' launch order, the
    ranks' sums added by the 256-thread tree (what tests/test_gpu_slab_plans.py holds the z-slab drivers to) -- with ONE
    rank must be the C twin (oracle_bicgstab_wr_gpuorder) bit for bit: x, iter, both histories; and with two and three
    ranks it must still be src/solvers.f90:3-50: the same solution to the solver tolerance, a true residual below it."""
    N = 16
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    n = N ** 3
    rng = np.random.Generator(np.random.PCG64(3))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    tol = 1e-9
    one = _FakeSlab(oracle, n, 8)
    g = oracle.geoms_of(one, (0,))[0]
    xc, itc, hsc, hrc = oracle.bicgstab_wr_gpuorder(g, valA, irow, jcol, b, x0, tol, 5000, hist_cap=32)
    xt, itt, hst, hrt, rst = oracle.twin_solve_slabs([(one, 0, n)], 0, valA, irow, jcol, b, x0, tol, 5000, hist_cap=32)
    assert itt == itc and np.array_equal(xt, xc)
    assert np.array_equal(hst[:min(itt, 32)], hsc[:min(itt, 32)]) and np.array_equal(hrt[:min(itt, 32) - 1], hrc[:min(itt, 32) - 1])
    assert rst == oracle.last_restart_count()
    for world in (2, 3):
        cuts = [n * r // world // 256 * 256 for r in range(world)] + [n]
        slabs = [(_FakeSlab(oracle, cuts[r + 1] - cuts[r], 8), cuts[r], cuts[r + 1]) for r in range(world)]
        xw, itw, _, _, _ = oracle.twin_solve_slabs(slabs, 0, valA, irow, jcol, b, x0, tol, 5000)
        res = np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, xw)) / np.linalg.norm(b)
        assert res < 5 * tol and np.linalg.norm(xw - xc) <= 1e-6 * np.linalg.norm(xc)
        assert abs(itw - itc) <= 0.25 * itc


def test_tree_sum_is_the_kernels_tree(oracle):
    """oracle_tree_sum: thread t of 256 adds values t, t + 256, ... in order; 64-lane shuffle tree per wave; the four wave
    sums left to right (reduce_partials / k_finalize in csrc/ec3d_kernels.hip).  Eight rank sums: ((v0+v4)+(v2+v6)) +
    ((v1+v5)+(v3+v7))."""
    v = np.array([1e16, 1.0, -1e16, 1.0, 3.0, 1e-3, 7.0, -1.0])
    want = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]))
    assert oracle.tree_sum(v) == want
    w = np.arange(1.0, 601.0)            # 600 values: threads 0..87 add three each, the others two
    acc = np.zeros(256)
    for i, x in enumerate(w):
        acc[i % 256] = acc[i % 256] + x
    waves = []
    for k in range(4):
        q = acc[64 * k:64 * k + 64].copy()
        off = 32
        while off:
            q[:off] = q[:off] + q[off:2 * off]
            off //= 2
        waves.append(q[0])
    assert oracle.tree_sum(w) == ((waves[0] + waves[1]) + waves[2]) + waves[3]
