"""Edge cases of the hot path on the GPU: ragged/tiny/odd grids, explicit zeros, unsorted and duplicate
CSR entries, degenerate iteration limits, the drop-in's matrix cache."""
import os

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


@pytest.mark.parametrize("dims,bnd", [((3, 3, 3), -0.95), ((7, 5, 4), -0.95), ((17, 9, 5), 0.0),
                                      ((33, 1 + 30, 3), 1.0)])
def test_odd_tiny_grids_and_explicit_zero_coefficients(E, oracle, dims, bnd):
    """sdx odd (the +-sdx pairs are only 8-byte aligned), n < one tile, BND = 0 (the reference then stores
    explicit zeros in its CSR; ours are band slots holding 0.0): SpMV and solve bit-identical."""
    sdx, sdy, sdz = dims
    delta = (0.004, 0.003, 0.005)
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz, delta, bnd)
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(n))
    x, b = rng.standard_normal(n), rng.standard_normal(n)
    for route in ("native", "csr"):
        with E.EC3DSolver() as s:
            if route == "native":
                s.assemble_poisson(sdx, sdy, sdz, delta, bnd)
            else:
                s.set_matrix_csr(valA, irow, jcol)
            assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))
            xs, it, _ = s.solve(b, np.zeros(n), 1e-10, 3000)
            xo, ito, _, _ = oracle.twin_solve(s, valA, irow, jcol, b, np.zeros(n),
                                                        1e-10, 3000)
            assert it == ito and np.array_equal(xs, xo)


def test_csr_stored_order_is_kept_for_unsorted_and_duplicate_entries(E, oracle):
    """src/solvers.f90:59 sums a row in STORED order; a CSR that is not column-sorted or repeats a column
    must give the same bits (such rows go to the tail as they are)."""
    N = 9
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    valA, jcol = valA.copy(), jcol.copy()
    rng = np.random.Generator(np.random.PCG64(17))
    for r in rng.choice(N ** 3, 200, replace=False):           # shuffle some rows
        p0, p1 = irow[r] - 1, irow[r + 1] - 1
        perm = rng.permutation(p1 - p0)
        valA[p0:p1], jcol[p0:p1] = valA[p0:p1][perm], jcol[p0:p1][perm]
    rows = [(jcol[irow[r] - 1:irow[r + 1] - 1], valA[irow[r] - 1:irow[r + 1] - 1]) for r in range(N ** 3)]
    for r in rng.choice(N ** 3, 50, replace=False):            # and repeat a column in others
        c, v = rows[r]
        rows[r] = (np.append(c, c[0]), np.append(v, 0.125))
    irow2 = np.concatenate([[1], 1 + np.cumsum([len(c) for c, _ in rows])]).astype(np.int32)
    jcol2 = np.concatenate([c for c, _ in rows]).astype(np.int32)
    valA2 = np.concatenate([v for _, v in rows])
    x = rng.standard_normal(N ** 3)
    with E.EC3DSolver() as s:
        s.set_matrix_csr(valA2, irow2, jcol2)
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(valA2, irow2, jcol2, x))


@pytest.mark.parametrize("itmax,expect", [(0, 1), (-1, 0), (-5, 0)])
def test_degenerate_iteration_limits(E, oracle, capfd, itmax, expect):
    """src/solvers.f90:25-29: the limit is tested before the increment: itmax = 0 -> one iteration,
    itmax < 0 -> none (x untouched), ||R|| printed in both cases."""
    N = 8
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    b = oracle.bar_rhs(N)
    x0 = np.full(N ** 3, 1e-3)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x, it, _ = s.solve(b, x0, 1e-30, itmax)
    xo, ito, _, _ = oracle.bicgstab_wr(valA, irow, jcol, b, x0, 1e-30, itmax)
    out = capfd.readouterr().out.split()
    assert it == ito == expect
    if expect == 0:
        assert np.array_equal(x, x0)
    else:
        assert np.allclose(x, xo, rtol=1e-12, atol=0)
    assert len(out) >= 2 and float(out[0]) == pytest.approx(float(out[1]), rel=1e-10)   # both printed ||R||


def test_dropin_cache_notices_a_matrix_rebuilt_in_place(E, oracle):
    """The device copy of the matrix is cached across calls (the reference assembles once); a host that
    rewrites valA in place is still served correctly (value signature), and ec3d_invalidate() exists."""
    N = 10
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    valA = valA.copy()
    b = oracle.bar_rhs(N)
    n = N ** 3
    x1 = np.zeros(n)
    it1 = E.sprsBCGstabWR(valA, irow, jcol, n, b, x1, 1e-9, 5000)
    valA *= 2.0                                   # same arrays, new operator: solution halves
    x2 = np.zeros(n)
    it2 = E.sprsBCGstabWR(valA, irow, jcol, n, b, x2, 1e-9, 5000)
    assert np.linalg.norm(2 * x2 - x1) <= 1e-6 * np.linalg.norm(x1)
    E.load_library().ec3d_invalidate()
    x3 = np.zeros(n)
    E.sprsBCGstabWR(valA, irow, jcol, n, b, x3, 1e-9, 5000)
    assert np.array_equal(x3, x2)


def test_nan_rhs_propagates_like_the_reference(E, oracle):
    """No breakdown guards in the reference (src/solvers.f90:32, :40, :45): NaNs propagate and the loop
    runs to itmax; same here (no hang, iter = itmax + 1)."""
    N = 6
    b = oracle.bar_rhs(N).copy()
    b[10] = np.nan
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x, it, _ = s.solve(b, np.zeros(N ** 3), 1e-6, 7)
    assert it == 8 and np.all(np.isnan(x[np.isfinite(x) == False])) and np.isnan(x).any()


@pytest.mark.parametrize("structured", [True, False])
@pytest.mark.parametrize("case", ["on_the_box", "two_cells_thin"])
def test_assembly_stops_where_the_reference_stops(E, oracle, case, structured):
    """Geometries the reference cannot assemble (it STOPs at src/EC3D.f90:717-720 / :945-948 or indexes out of
    range): the device assembly refuses them with the status the oracle's restatement reports, in both storage
    formats, and the handle stays usable."""
    g = load_golden("g2_conducting_hole_16x15x14")
    geo, geoC = g["geoPHYS"].copy(), np.zeros_like(g["geoPHYS_C"])
    cond = np.zeros(geo.shape, bool)
    if case == "on_the_box":
        cond[0:4, 4:9, 4:9] = True          # touches the z = 1 face
    else:
        cond[5:7, 4:9, 4:9] = True          # only two cells thick in z: the one-sided stencil has no third cell
    geo[cond] = 1
    geo[~cond & (geo == 1)] = geo.max()
    n_cells = geo.size
    geoC[cond] = 3 * n_cells + 1 + np.arange(int(cond.sum()))
    args = (geo, geoC, g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
    with pytest.raises(RuntimeError, match="STOP") as ref:
        oracle.gen_sparse_matrix(*args)
    code = int(str(ref.value).split("code ")[1].rstrip(")"))
    with E.EC3DSolver(structured=structured) as s:
        with pytest.raises(E.EC3DError) as err:
            s.assemble(*args)
        assert f"({code})" in str(err.value), (str(err.value), code)
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))   # still usable
        assert s.n == len(g["irow"]) - 1


def test_parked_band_placement_is_reused_once_and_released(E):
    """Plain band streams of >= 32 Mi rows get a placement probe at set-up (place_bands); the chosen
    allocation is parked when the matrix is replaced and handed back to the next matrix of the same size (no second probe,
    no transient second copy), and released as soon as a matrix arrives that does not take it (another size, another
    storage).  A*x and a short solve on the re-assembled handle equal a fresh handle's bit for bit; the device memory
    the parked copy held (3.7 GB here) is free again afterwards."""
    import torch
    sdx, sdy, sdz = 512, 512, 128
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(31))
    x = rng.standard_normal(n)
    b = rng.standard_normal(n)
    with E.EC3DSolver(dictionary=False) as fresh:
        fresh.assemble_poisson(sdx, sdy, sdz)
        y_ref = fresh.spmv(x)
        x_ref, it_ref, _ = fresh.solve(b, np.zeros(n), 1e-30, 4)
    with E.EC3DSolver(dictionary=False) as s:
        s.assemble_poisson(sdx, sdy, sdz)
        first = s.band_placement()
        assert len(first[0]) >= 1 and 0 <= first[1] < len(first[0])
        s.assemble_poisson(sdx, sdy, sdz)                    # the same size again: the parked allocation comes back
        assert s.band_placement() == first                   # no new probe
        assert np.array_equal(s.spmv(x), y_ref)
        x2, it2, _ = s.solve(b, np.zeros(n), 1e-30, 4)
        assert it2 == it_ref == 5 and np.array_equal(x2, x_ref)
        torch.cuda.synchronize()
        free_with_bands = torch.cuda.mem_get_info()[0]
        s.assemble_poisson(256, 256, 64)                     # another size: nothing parked may survive
        xs = rng.standard_normal(256 * 256 * 64)
        with E.EC3DSolver(dictionary=False) as small:
            small.assemble_poisson(256, 256, 64)
            assert np.array_equal(s.spmv(xs), small.spmv(xs))
        torch.cuda.synchronize()
        free_small = torch.cuda.mem_get_info()[0]
        # the large system held 7 band streams + 8 work vectors + 7 ring buffers of 8 n bytes each: all of it must be free
        # again; a parked copy of the band streams (7 of the 22) would leave the difference at 15
        assert free_small - free_with_bands > 8 * n * 18.5


def test_iterate_must_continue_its_numbering(E):
    """ec3d_iterate addresses the device state by the iteration number (rr0[it & 1], AP in apbuf[it & 1], P and S in rings of
    up to four buffers): a call that does not continue where the last one ended -- iterate(1, 5) twice -- would iterate on
    an older P and the other rr0, so it is refused (status 6); pieces that do continue give what one call gives; and
    ec3d_iterate_begin starts the numbering again."""
    import os
    N = 32
    n = N ** 3
    b = np.random.Generator(np.random.PCG64(12)).standard_normal(n)
    os.environ["EC3D_XDEFER"] = "4"            # rings of four P / S buffers on this small grid too
    try:
        outs = []
        for pieces in ((9,), (5, 4), (2, 3, 4)):
            with E.EC3DSolver() as s:
                s.assemble_poisson(N, N, N)
                assert s.x_interval() == 4
                s.upload("B", b)
                s.upload("X", np.zeros(n))
                s.iterate_begin()
                at = 1
                for c in pieces:
                    s.iterate(at, c)
                    at += c
                s.synchronize()
                outs.append(s.download("X"))
                with pytest.raises(E.EC3DError, match="does not continue"):
                    s.iterate(1, 5)
                s.iterate_begin()
                s.iterate(1, 3)
                s.synchronize()
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    finally:
        del os.environ["EC3D_XDEFER"]


def test_vector_placement_search_leaves_no_trace_and_is_kept(E):
    """Work vectors of >= 32 Mi rows get a placement search at set-up (place_vectors): a right-hand side
    of ones is iterated on up to EC3D_PLACE_VEC allocations of vectors + rings and the fastest kept.  Whatever it picks,
    nothing of it may show: a solve afterwards equals a handle's that never searched, bit for bit (x, iteration count,
    residual history); the chosen allocation is kept for the next matrix of the same size (no second search).  Forced at
    small sizes through ec3d_place_vectors -- the five-launch iteration on a cube and the structured A-V system -- and by
    itself on 512 x 512 x 128 (the three-launch iteration with the X update deferred)."""
    rng = np.random.Generator(np.random.PCG64(77))
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g2_conducting_hole_16x15x14.npz"))

    def cube(s):
        s.assemble_poisson(128, 128, 48)

    def av(s):
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))

    for setup, tol, itmax in ((cube, 1e-9, 60), (av, float(g["tol"]), int(g["itmax"]))):
        with E.EC3DSolver() as fresh:
            setup(fresh)
            n = fresh.n
            b = rng.standard_normal(n)
            assert fresh.vector_placement() == ([], -1, 0.0)         # below 32 Mi rows nothing is searched
            x_ref, it_ref, h_ref = fresh.solve(b, np.zeros(n), tol, itmax, hist_cap=64)
        with E.EC3DSolver() as s:
            setup(s)
            us, kept, ms = s.place_vectors(3)
            assert 2 <= len(us) <= 3 and 0 <= kept < len(us) and all(u > 0 for u in us) and ms > 0
            x, it, h = s.solve(b, np.zeros(n), tol, itmax, hist_cap=64)
            assert it == it_ref and np.array_equal(x, x_ref) and np.array_equal(h, h_ref, equal_nan=True)
            setup(s)                                                 # the same size again: allocation kept, no search
            assert s.vector_placement() == (us, kept, ms)
            x, it, h = s.solve(b, np.zeros(n), tol, itmax, hist_cap=64)
            assert it == it_ref and np.array_equal(x, x_ref) and np.array_equal(h, h_ref, equal_nan=True)
    sdx, sdy, sdz = 512, 512, 128
    n = sdx * sdy * sdz
    b = rng.standard_normal(n)
    os.environ["EC3D_PLACE_VEC"] = "0"
    try:
        with E.EC3DSolver() as plain:
            plain.assemble_poisson(sdx, sdy, sdz)
            assert plain.vector_placement() == ([], -1, 0.0)
            x_ref, it_ref, h_ref = plain.solve(b, np.zeros(n), 1e-30, 9, hist_cap=16)
    finally:
        del os.environ["EC3D_PLACE_VEC"]
    with E.EC3DSolver() as s:
        s.assemble_poisson(sdx, sdy, sdz)
        us, kept, ms = s.vector_placement()
        assert 2 <= len(us) <= 6 and 0 <= kept < len(us) and us[kept] == min(us) and ms < 20000   # (a hipMalloc now and then takes seconds on a box that has just released memory)
        x, it, h = s.solve(b, np.zeros(n), 1e-30, 9, hist_cap=16)
        assert it == it_ref == 10 and np.array_equal(x, x_ref) and np.array_equal(h, h_ref, equal_nan=True)
