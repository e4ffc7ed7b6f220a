"""The schedules of the z-slab drivers against the oracle's MULTI-RANK twin, bit for bit.

The multi-GPU handle (include/ec3d_hip.h section 2c, csrc/ec3d_multi.hip; on the one-GPU test box every slab sits on
device 0: the same code with local copies instead of xGMI ones) runs one of four plans (ec3d_multi_plan):

  0  five launches, halo exchange in front of K1 and K3
  1  K1 / K3 as interior + boundary launch, the exchange behind the interior one
  2  K2 / K5 boundary tiles first (A-V slabs; pinned elsewhere against the staged driver and the reference's captures)
  3  three launches per iteration -- K2 inside K3, K4 as an SpMV kernel that computes A S again, K5 inside the next K1 --
     with AP and R exchanged instead of P and S, S and P formed on the halo planes by the kernels that read them there
  4  plan 3 with the producers of R and AP (K4, K5-in-K1) as a boundary launch (planes 0 and np-1) and an interior launch,
     the exchange behind the boundary launch
  5  plans 1 and 2 together (EC3D_SLAB_PLAN=5): K2 / K5 boundary planes first AND K1 / K3 interior planes first -- the
     exchange behind two launches

and, on every plan, X = X + alpha*P + omega*S (src/solvers.f90:41) applied every D-th iteration from rings of P and S.
oracle.twin_solve_slabs restates src/solvers.f90:3-50 on the WHOLE system and sums every dot product the way the slabs
do: per rank in the order of that rank's launches (ec3d_get_visit_order, a split kernel's partials strung together),
collapsed by the 256-thread tree, the ranks' sums added in rank order.  x, the iteration count and the number of
restarts (:47-49) must be the twin's.  Sizes are small and the size policies forced through the environment
(EC3D_FUSE23 / EC3D_FUSE51 / EC3D_K4S = 2, EC3D_XDEFER, EC3D_NT); tests/test_gpu_config4.py runs the plans the library
picks by itself at the slab shapes of the 512^3 cube."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KNOBS = ("EC3D_SLAB_FSPLIT", "EC3D_SLAB_PLAN", "EC3D_NT", "EC3D_KEEP", "EC3D_FUSE23", "EC3D_FUSE51", "EC3D_PATCH", "EC3D_NBLK", "EC3D_NBLK_SPMV", "EC3D_VEC_DEPTH",
         "EC3D_XCD_MAP", "EC3D_ZMARCH", "EC3D_XDEFER", "EC3D_XD_OFF_DEPTH", "EC3D_XD_ON_DEPTH", "EC3D_K4S", "EC3D_SLAB_FUSE",
         "EC3D_SLAB_XDEFER", "EC3D_XASYNC", "EC3D_XASYNC_WGS", "EC3D_XASYNC_PRIO")


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


def set_knobs(monkeypatch, **kw):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in kw.items():
        monkeypatch.setenv("EC3D_" + k, str(v))


def slabs_of(m, kdz):
    out = []
    for r in range(m.nranks):
        view, k0, k1 = m.slab(r)
        out.append((view, k0 * kdz, k1 * kdz))
    return out


def restarts_of(m):
    return [m.slab(r)[0].restart_count() for r in range(m.nranks)]


FUSED = dict(FUSE23=2, FUSE51=2, K4S=2)


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("xd", [1, 4])
@pytest.mark.parametrize("nt, fsplit", [(0, 1), (1, 1), (0, 0)])
def test_three_launch_iteration_on_slabs_bitwise(E, oracle, monkeypatch, world, xd, nt, fsplit):
    """Plans 3 and 4 on 2, 3 and 4 slabs (uneven cuts included: 50 planes over 3 and 4 ranks), with the X update in every
    iteration and every fourth, cacheable and nontemporal streams: converged solves with both exits available."""
    sdx, sdy, sdz = 128, 8, 50
    n, kdz = sdx * sdy * sdz, sdx * sdy
    set_knobs(monkeypatch, XDEFER=xd, NT=nt, SLAB_FSPLIT=fsplit, **FUSED)
    P3 = 4 if fsplit else 3
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(500 + world))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    tol = 1e-6
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        assert m.plan() == (P3, xd)
        v0 = m.slab(0)[0]
        assert v0.fusion() == (1, 1) and v0.k4_as_spmv() and v0.x_interval() == xd
        assert np.array_equal(m.spmv(x0), oracle.spmv_csr(valA, irow, jcol, x0))      # (the probe's exchange: plain vectors)
        x, it = m.solve(b, x0, tol, 5000)
        rs = restarts_of(m)
        xo, ito, _, _, rso = oracle.twin_solve_slabs(slabs_of(m, kdz), P3, valA, irow, jcol, b, x0, tol, 5000)
        # a second solve on the same handle, warm-started from the first one's solution
        x2, it2 = m.solve(b, x, tol, 5000)
        xo2, ito2, _, _, _ = oracle.twin_solve_slabs(slabs_of(m, kdz), P3, valA, irow, jcol, b, xo, tol, 5000)
    res = np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b)
    print(f"{world} slabs, plan {P3}, X every {xd}, nt {nt}: iter {it} (twin {ito}), restarts {rs[0]} (twin {rso}), "
          f"true residual {res:.2e}")
    assert it == ito and np.array_equal(x, xo)
    assert all(r == rso for r in rs)
    assert it2 == ito2 and np.array_equal(x2, xo2)
    assert res < 5 * tol


@pytest.mark.parametrize("fused", [True, False], ids=["three-launches", "five-launches"])
def test_restart_rule_on_slabs(E, oracle, monkeypatch, fused):
    """The restart R0 = R, P = R (src/solvers.f90:47-49) on slabs: inside K5-in-K1 it also decides what the kernel forms on
    the halo planes (P = R there too).  Tolerance and start as in tests/test_gpu_parity.py's 2-D-tile case, where the rule
    fires; counted on every slab's device and in the twin."""
    sdx, sdy, sdz = 256, 8, 31
    n, kdz = sdx * sdy * sdz, sdx * sdy
    set_knobs(monkeypatch, XDEFER=4, **(FUSED if fused else dict(SLAB_PLAN=5)))
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(2026))
    x0 = np.zeros(n)
    rng.standard_normal(n)
    b = rng.standard_normal(n)
    tol = 1e-9
    with E.EC3DMulti(3, devices=[0, 0, 0]) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        plan = m.plan()[0]
        assert plan == (4 if fused else 5)     # (five launches: plan 5 -- both splits -- asked for; opt-in since round 6)
        x, it = m.solve(b, x0, tol, 5000)
        rs = restarts_of(m)
        xo, ito, _, _, rso = oracle.twin_solve_slabs(slabs_of(m, kdz), plan, valA, irow, jcol, b, x0, tol, 5000)
    print(f"plan {plan}: iter {it} (twin {ito}), restarts {rs} (twin {rso})")
    assert it == ito and np.array_equal(x, xo)
    assert rso > 0 and all(r == rso for r in rs)


@pytest.mark.parametrize("plan, dims", [(1, (128, 8, 48)), (0, (24, 24, 24)), (2, (128, 8, 48)), (0, (128, 8, 48)), (5, (128, 8, 48))],
                         ids=["interior+boundary", "plain", "producers-split", "plain-zmarch", "both-split"])
@pytest.mark.parametrize("xd", [1, 3, 4])
def test_five_launch_plans_with_deferred_x_bitwise(E, oracle, monkeypatch, plan, dims, xd):
    """Plans 0 and 1 (what the slabs of 512^3 on 8 GPUs run: 16 Mi rows per rank) with the X update every D-th iteration:
    P and S live in rings and the halo exchange follows them."""
    sdx, sdy, sdz = dims
    n, kdz = sdx * sdy * sdz, sdx * sdy
    world = 3
    set_knobs(monkeypatch, XDEFER=xd, SLAB_PLAN=plan)
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(77))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    tol = 1e-7
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        assert m.plan() == (plan, xd)
        x, it = m.solve(b, x0, tol, 5000)
        rs = restarts_of(m)
        xo, ito, _, _, rso = oracle.twin_solve_slabs(slabs_of(m, kdz), plan, valA, irow, jcol, b, x0, tol, 5000)
    print(f"plan {plan}, X every {xd}: iter {it} (twin {ito}), restarts {rs[0]} (twin {rso})")
    assert it == ito and np.array_equal(x, xo)
    assert all(r == rso for r in rs)


@pytest.mark.parametrize("fused", [True, False], ids=["three-launches", "five-launches"])
def test_itmax_exit_at_every_position_of_a_group(E, oracle, monkeypatch, fused):
    """src/solvers.f90:25-29 ends the loop after itmax + 1 iterations; with X applied every fourth iteration the last
    iteration of the call applies whatever is pending.  k = 1 .. 9 iterations: every position of a group of four."""
    sdx, sdy, sdz = 128, 8, 32
    n, kdz = sdx * sdy * sdz, sdx * sdy
    set_knobs(monkeypatch, XDEFER=4, **(FUSED if fused else dict(SLAB_PLAN=5)))
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(9))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        plan = m.plan()[0]
        assert plan == (4 if fused else 5)     # (five launches: plan 5 -- both splits -- asked for; opt-in since round 6)
        for k in range(1, 10):
            x, it = m.solve(b, x0, 1e-30, k - 1)
            xo, ito, _, _, _ = oracle.twin_solve_slabs(slabs_of(m, kdz), plan, valA, irow, jcol, b, x0, 1e-30, k - 1)
            assert it == ito == k and np.array_equal(x, xo), k


def test_s_exit_with_updates_pending(E, oracle, monkeypatch):
    """The ||S|| exit (src/solvers.f90:34-38: X = X + alpha*P) taken in the middle of a group: a right-hand side the first
    half step solves exactly enough -- b = A*(x0 + c*r0-direction) is hard to hit, so the tolerance is chosen between the
    ||S|| and ||R|| values of an iteration instead: the exit must be the twin's, whichever it is, with X complete."""
    sdx, sdy, sdz = 128, 8, 32
    n, kdz = sdx * sdy * sdz, sdx * sdy
    set_knobs(monkeypatch, XDEFER=4, **FUSED)
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(10))
    b = rng.standard_normal(n)
    x0 = np.zeros(n)
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        P3 = m.plan()[0]
        assert P3 == 4
        _, _, hs, hr, _ = oracle.twin_solve_slabs(slabs_of(m, kdz), P3, valA, irow, jcol, b, x0, 1e-30, 13, hist_cap=14)
        bn = np.linalg.norm(b)
        seen = set()
        for k in range(2, 13):
            # between ||S_k|| and ||R_{k-1}||: iteration k leaves by its ||S|| test when S is the smaller one
            lo, hi = sorted((hs[k - 1], hr[k - 2]))
            tol = 0.5 * (lo + hi) / bn
            x, it = m.solve(b, x0, tol, 100)
            xo, ito, _, _, _ = oracle.twin_solve_slabs(slabs_of(m, kdz), P3, valA, irow, jcol, b, x0, tol, 100)
            assert it == ito and np.array_equal(x, xo), k
            seen.add((it - 1) % 4)
    assert len(seen) >= 3       # exits at several positions of a group of four


def test_iterate_continues_and_refuses_to_restart_its_numbering(E, oracle, monkeypatch):
    """ec3d_multi_iterate (bench.py's timed region) in pieces: the device state is addressed by the iteration number
    (rr0[it & 1], AP in apbuf[it & 1], P and S in their rings), so iterate(1, 5) + iterate(6, 5) == iterate(1, 10), and a
    call that does not continue the numbering -- iterate(1, 5) twice -- is refused instead of reading an older P."""
    sdx, sdy, sdz = 128, 8, 32
    n = sdx * sdy * sdz
    set_knobs(monkeypatch, XDEFER=4, **FUSED)
    rng = np.random.Generator(np.random.PCG64(11))
    b = rng.standard_normal(n)
    out = []
    for pieces in ((10,), (5, 5), (3, 3, 4)):
        with E.EC3DMulti(2, devices=[0, 0]) as m:
            m.assemble_poisson(sdx, sdy, sdz)
            m.upload("B", b)
            m.upload("X", np.zeros(n))
            m.iterate_begin()
            at = 1
            for c in pieces:
                m.iterate(at, c)
                at += c
            m.synchronize()
            out.append(m.download("X"))
            with pytest.raises(E.EC3DError, match="does not continue"):
                m.iterate(1, 5)
            m.iterate_begin()          # ... which starts again from 1
            m.iterate(1, 2)
            m.synchronize()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[2])


@pytest.mark.parametrize("fused", [True, False], ids=["three-launches", "five-launches"])
def test_x_groups_on_a_second_stream_bitwise(E, oracle, monkeypatch, fused):
    """EC3D_XASYNC: no K4 touches X; every group of four updates is applied by a launch of its own (k_x_group) on a second
    stream beside the iterations that follow, P and S in rings of two groups, alpha / omega in entry it % 8 of the device
    state.  The same additions in the same order: x, iter and restarts are the twin's -- on the system where the restart
    rule fires, with the itmax exit at every position of a group (the last group of a call is cut short and joined), and
    with the ||S|| exit inside a group (the group's launch, or the flush the host adds when it had not enqueued it yet,
    ends at the exit with the half update).  The second launch runs on 8 workgroups here, so its stride differs from
    the vector kernels' grid."""
    set_knobs(monkeypatch, XDEFER=4, XASYNC=2, XASYNC_WGS=8, **(FUSED if fused else dict(SLAB_PLAN=5)))
    # -- restarts, a full solve, three slabs
    sdx, sdy, sdz = 256, 8, 31
    n, kdz = sdx * sdy * sdz, sdx * sdy
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(2026))
    x0 = np.zeros(n)
    rng.standard_normal(n)
    b = rng.standard_normal(n)
    with E.EC3DMulti(3, devices=[0, 0, 0]) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        plan = m.plan()[0]
        assert plan == (4 if fused else 5)     # (five launches: plan 5 -- both splits -- asked for; opt-in since round 6)
        x, it = m.solve(b, x0, 1e-9, 5000)
        rs = restarts_of(m)
        on = [m.slab(r)[0].x_groups() for r in range(3)]
        xo, ito, _, _, rso = oracle.twin_solve_slabs(slabs_of(m, kdz), plan, valA, irow, jcol, b, x0, 1e-9, 5000)
    assert all(a and g >= it // 4 for a, g in on), on          # the mode was on, and a launch per group went out
    assert it == ito and np.array_equal(x, xo) and rso > 0 and all(r == rso for r in rs)
    # -- itmax at every position of a group; the ||S|| exit inside a group; two slabs
    sdx, sdy, sdz = 128, 8, 32
    n, kdz = sdx * sdy * sdz, sdx * sdy
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    rng = np.random.Generator(np.random.PCG64(10))
    b = rng.standard_normal(n)
    x0 = np.zeros(n)
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        plan = m.plan()[0]
        sl = slabs_of(m, kdz)
        for k in range(1, 10):
            x, it = m.solve(b, x0, 1e-30, k - 1)
            xo, ito, _, _, _ = oracle.twin_solve_slabs(sl, plan, valA, irow, jcol, b, x0, 1e-30, k - 1)
            assert it == ito == k and np.array_equal(x, xo), k
        _, _, hs, hr, _ = oracle.twin_solve_slabs(sl, plan, valA, irow, jcol, b, x0, 1e-30, 13, hist_cap=14)
        bn = np.linalg.norm(b)
        seen = set()
        for k in range(2, 13):
            lo, hi = sorted((hs[k - 1], hr[k - 2]))
            tol = 0.5 * (lo + hi) / bn
            x, it = m.solve(b, x0, tol, 100)
            xo, ito, _, _, _ = oracle.twin_solve_slabs(sl, plan, valA, irow, jcol, b, x0, tol, 100)
            assert it == ito and np.array_equal(x, xo), k
            seen.add((it - 1) % 4)
        assert m.slab(0)[0].x_groups()[0]
    assert len(seen) >= 3
