"""BASELINE config 4 -- the synthetic 512^3 7-point operator cut into 8 z-slabs -- as far as one card can take it.

SURVEY section 8d "Config 4": n = 134 217 728, nnz = 937 951 232, z-slabs of 64 planes, the reference solver timed
for a fixed 20 iterations only.  The workload bench.py quotes its number on is exercised here at its full size:

* A*x of the undivided handle (src/solvers.f90:54-61 on the src/EC3D.f90:528-654 operator) against the oracle's CSR
  row sums, bit for bit, on the planes where something could go wrong (box faces, the seven slab cuts of the 8-way
  split and the planes either side of them, the middle): the CSR triple of the whole cube is 11 GB, so the oracle
  builds the CSR of those planes only (global columns) and sums them against the whole vector;
* A*x over 8 slabs on this one card (slab operators + halo planes pulled between the slabs) == the undivided
  handle, bit for bit, all 134 M rows;
* 20 fixed iterations (src/solvers.f90:24-50, exits disabled by tol = 1e-300, itmax = 19: the reference's
  "iter > itmax" test before the increment runs exactly 20) on 8 slabs inside the library == the staged driver of
  eddy_currents_3d_amd/dist.py on the same 8 slabs, bit for bit, and within rounding growth of the undivided
  handle's 20 iterations (a different summation tree for the dot products, nothing else)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, WORLD = 512, 8


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    return E


@pytest.fixture(scope="module")
def xvec():
    return np.random.Generator(np.random.PCG64(404)).standard_normal(N ** 3)


@pytest.fixture(scope="module")
def undivided(E, xvec):
    """y = A x and 20 iterations from (bar RHS, x0 = 0) on the undivided 512^3 handle."""
    from bench import bar_rhs
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        assert s.n == N ** 3 and s.info.nnz == 7 * N ** 3 - 6 * N * N          # 937 951 232
        y = s.spmv(xvec)
        x20, it, _ = s.solve(bar_rhs(N), np.zeros(N ** 3), 1e-300, 19)
    assert it == 20
    return y, x20


def test_x_every_fourth_iteration_leaves_the_same_bits_at_full_size(E, undivided, monkeypatch):
    """The headline handle applies X = X + alpha*P + omega*S (src/solvers.f90:41) every fourth iteration (k4s_x_r_spmv /
    k4d_x_r_update, rings of P and S).  A size-independent property: the same 20 iterations with an update in every
    iteration (EC3D_XDEFER=1, everything else as the library picks it) give the same x, all 134 M entries, bit for bit."""
    from bench import bar_rhs
    _, x20 = undivided
    monkeypatch.setenv("EC3D_XDEFER", "1")
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        assert s.x_interval() == 1 and s.fusion() == (1, 1) and s.k4_as_spmv()
        x1, it, _ = s.solve(bar_rhs(N), np.zeros(N ** 3), 1e-300, 19)
    monkeypatch.delenv("EC3D_XDEFER")
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        assert s.x_interval() == 4                                   # what `undivided` ran with
    assert it == 20 and np.array_equal(x1, x20)


def test_undivided_rows_equal_the_oracle_csr_row_sums(oracle, xvec, undivided):
    y, _ = undivided
    kd = N * N
    cuts = sorted({0, 1, N // 2 - 1, N // 2, N - 2, N - 1} |
                  {c + d for c in range(N // WORLD, N, N // WORLD) for d in (-1, 0)})
    checked = 0
    for k in cuts:
        want = oracle.poisson_rows_times(N, N, N, k, k + 1, xvec)
        assert np.array_equal(y[k * kd:(k + 1) * kd], want), f"plane {k}"
        checked += kd
    print(f"512^3: {checked} rows on {len(cuts)} planes bit-identical to the oracle's CSR row sums")


def test_plain_band_streams_equal_the_dictionary_form(E, xvec, undivided, capfd):
    """The north-star SpMV figure's storage (seven fp64 coefficient streams, 7.5 GB at this size) multiplies the same
    doubles in the same order as the class-coded form, wherever the library's set-up probe (place_bands: up to 8
    placements of the streams timed, the fastest kept) ends up putting them."""
    import os
    y, _ = undivided
    os.environ["EC3D_PLACE_VERBOSE"] = "1"
    try:
        with E.EC3DSolver(dictionary=False) as s:
            s.assemble_poisson(N, N, N)
            yd = s.spmv(xvec)
    finally:
        del os.environ["EC3D_PLACE_VERBOSE"]
    err = capfd.readouterr().err
    tried = [l for l in err.splitlines() if "band placement" in l]
    print("\n".join(tried))
    assert 1 <= len(tried) <= 8
    assert np.array_equal(yd, y)


def test_eight_slabs_on_one_card_spmv_equals_undivided(E, xvec, undivided):
    y, _ = undivided
    with E.EC3DMulti(WORLD, devices=[0] * WORLD) as m:
        m.assemble_poisson(N, N, N)
        assert m.n == N ** 3
        assert all(m.slab(r)[0].can_overlap() for r in range(WORLD))
        y8 = m.spmv(xvec)
    assert np.array_equal(y8, y)


def test_twenty_iterations_on_eight_slabs_equal_the_staged_driver(E, undivided):
    from bench import bar_rhs
    from eddy_currents_3d_amd.dist import HipSlabOps, InProcessSlabs, slab_bounds
    _, x1 = undivided
    b = bar_rhs(N)
    with E.EC3DMulti(WORLD, devices=[0] * WORLD) as m:
        m.assemble_poisson(N, N, N)
        x8, it8 = m.solve(b, np.zeros(N ** 3), 1e-300, 19)
    assert it8 == 20
    ops = []
    for r in range(WORLD):
        k0, k1 = slab_bounds(N, r, WORLD)
        o = HipSlabOps(N, N, N, k0, k1, WORLD)
        o.set_vector("B", b.reshape(N, N * N)[k0:k1].reshape(-1))
        ops.append(o)
    drv = InProcessSlabs(ops)
    it_s = drv.solve(1e-300, 19)
    xs = drv.x()
    for o in ops:
        o.close()
    assert it_s == 20 and np.array_equal(x8, xs)
    rel = float(np.linalg.norm(x8 - x1) / np.linalg.norm(x1))
    print(f"512^3 on 8 slabs, 20 iterations: bit-identical to the staged driver; ||x_8slabs - x_undivided|| / ||x|| = {rel:.2e}")
    assert rel <= 1e-9
