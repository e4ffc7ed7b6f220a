"""SEVERAL ranks of the one-process-per-GPU driver (ec3d_multi_create_rank, csrc/ec3d_multi.hip) on ONE GPU.

RCCL refuses two ranks on one device (tools/rccl_two_ranks_one_gpu_probe.py), so tests/test_gpu_rccl_rank.py can run that
driver only as a one-rank job or as the rehearsal of one rank.  What several ranks add -- which planes go to which
neighbour and in which order the sends and receives of a group pair up, where rank r's eight sums land in the gathered
table, the facts the ranks agree on at set-up (one plan, one X interval, one ring depth for the whole job), every rank
leaving the loop at the same iteration -- is covered here through the LOOPBACK transport (tests/support/rccl_loopback.cpp,
built into tests/libec3d_loopback.so -- NOT part of the product library -- and named in EC3D_RCCL_LIB, where a librccl
would be named): the RCCL entry points served by threads of this process that copy between the ranks'
buffers with the ordering the real calls give.  Every rank is a thread with its own handle, created exactly as a process
of the launcher's job creates it; everything above the transport is the product's code.

The bar: x (each rank returns its own planes), the iteration count, the plan, the true residual and A*x of the rank job
equal those of the ONE-process handle on the same slabs bit for bit -- and that handle is pinned against the oracle's
multi-rank twin in tests/test_gpu_slab_plans.py and against the reference's captures in tests/test_gpu_multi.py."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LOOPBACK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libec3d_loopback.so")

KNOBS = ("EC3D_SLAB_FSPLIT", "EC3D_SLAB_PLAN", "EC3D_NT", "EC3D_FUSE23", "EC3D_FUSE51", "EC3D_XDEFER", "EC3D_K4S", "EC3D_SLAB_FUSE",
         "EC3D_SLAB_XDEFER", "EC3D_XASYNC", "EC3D_XASYNC_WGS")


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
    if not os.path.exists(LOOPBACK):
        from eddy_currents_3d_amd import build
        build.build_test_support()
    monkeypatch.setenv("EC3D_RCCL_LIB", LOOPBACK)


def run_ranks(E, world, body, **fmt):
    """body(m, rank) on `world` threads, each with the handle of one rank of a loopback job; the list of results."""
    ids = [E.EC3DMulti.rccl_unique_id(), E.EC3DMulti.rccl_unique_id()]
    assert ids[0] != ids[1]
    out, err = [None] * world, [None] * world

    def rank_main(r):
        try:
            with E.EC3DMulti.for_rank(r, world, 0, ids[0], ids[1], **fmt) as m:
                out[r] = body(m, r)
        except BaseException as e:   # noqa: BLE001 -- reported by the caller's thread
            err[r] = e

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r, e in enumerate(err):
        if e is not None:
            raise AssertionError(f"rank {r}: {e!r}") from e
    return out


def merge(x0, parts):
    """Every rank returns the global vector with ITS rows filled (the others as handed in)."""
    x = x0.copy()
    for p in parts:
        mine = p != x0
        x[mine] = p[mine]
    return x


FUSED = dict(FUSE23=2, FUSE51=2, K4S=2)
CASES = [
    ("five-launches", (20, 12, 31), dict(), 0),
    ("interior+boundary", (128, 8, 48), dict(SLAB_PLAN=1), 1),
    ("interior+boundary by default", (128, 8, 48), dict(), 1),      # (plan 5 is opt-in until two real devices have verified it)
    ("both-split", (128, 8, 48), dict(SLAB_PLAN=5), 5),
    ("three-launches,X/4", (128, 8, 48), dict(FUSED, XDEFER=4, SLAB_FSPLIT=0), 3),
    ("three-launches-split,X/4", (128, 8, 48), dict(FUSED, XDEFER=4), 4),
    ("three-launches-split,X/3", (128, 8, 48), dict(FUSED, XDEFER=3), 4),
    ("both-split,X/4", (128, 8, 48), dict(XDEFER=4, SLAB_PLAN=5), 5),
    ("producers-split,X/4", (128, 8, 48), dict(XDEFER=4, SLAB_PLAN=2), 2),
    # the groups of X updates as launches of their own on a second stream (rings of two groups: the exchanged P and S live there)
    ("interior+boundary,X/4 beside the iteration", (128, 8, 48), dict(XDEFER=4, XASYNC=1, XASYNC_WGS=8, SLAB_PLAN=1), 1),
    ("both-split,X/4 beside the iteration", (128, 8, 48), dict(XDEFER=4, XASYNC=1, XASYNC_WGS=8, SLAB_PLAN=5), 5),
    ("three-launches-split,X/4 beside the iteration", (128, 8, 48), dict(FUSED, XDEFER=4, XASYNC=2, XASYNC_WGS=8), 4),
]


@pytest.mark.parametrize("world", [2, 3, 4])
@pytest.mark.parametrize("name, dims, knobs, plan", CASES, ids=[c[0] for c in CASES])
def test_rank_job_equals_the_one_process_handle(E, oracle, monkeypatch, world, name, dims, knobs, plan):
    set_knobs(monkeypatch, **knobs)
    sdx, sdy, sdz = dims
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(11))
    b = rng.standard_normal(n)
    xs = rng.standard_normal(n)
    x0 = np.zeros(n)
    tol, itmax = 1e-9, 400
    with E.EC3DMulti(world, devices=[0] * world) as one:
        one.assemble_poisson(sdx, sdy, sdz)
        want_plan = one.plan()
        x_one, it_one = one.solve(b, x0, tol, itmax)
        res_one = one.true_residual()
        y_one = one.spmv(xs)
        cuts = [one.slab(r)[1:] for r in range(world)]
    if sdz // world >= 10:  # (thinner slabs have no interior launch: the library falls back by itself, on both drivers)
        assert want_plan[0] == plan, want_plan

    def body(m, r):
        m.assemble_poisson(sdx, sdy, sdz)
        assert m.n == n and m.slab(0)[1:] == cuts[r] and m.plan() == want_plan
        x, it = m.solve(b, x0, tol, itmax)
        res = m.true_residual()
        y = m.spmv(xs)
        return x, it, res, y

    got = run_ranks(E, world, body)
    assert [g[1] for g in got] == [it_one] * world                 # every rank leaves at the same iteration
    assert all(g[2] == res_one for g in got)                       # ... and holds the same sums
    assert np.array_equal(merge(x0, [g[0] for g in got]), x_one)
    assert np.array_equal(merge(np.zeros(n), [g[3] for g in got]), y_one)
    assert it_one > 20


def test_restarts_and_itmax_on_a_rank_job(E, oracle, monkeypatch):
    """The system of tests/test_gpu_slab_plans.py::test_restart_rule_on_slabs (the rule of src/solvers.f90:47-49 fires) on
    three ranks, three launches per iteration, X every fourth: every rank restarts at the same iterations.  Then the same
    system stopped by itmax in the middle of a group of four (src/solvers.f90:59-61): every rank runs exactly itmax
    iterations (and reports itmax + 1, as the reference's counter stands then) and X holds all of them."""
    set_knobs(monkeypatch, **dict(FUSED, XDEFER=4))
    sdx, sdy, sdz, world = 256, 8, 31, 3
    n = sdx * sdy * sdz
    rng = np.random.Generator(np.random.PCG64(2026))
    x0 = np.zeros(n)
    rng.standard_normal(n)
    b = rng.standard_normal(n)
    with E.EC3DMulti(world, devices=[0] * world) as one:
        one.assemble_poisson(sdx, sdy, sdz)
        x_one, it_one = one.solve(b, x0, 1e-9, 5000)
        restarts = [one.slab(r)[0].restart_count() for r in range(world)]
        x_cut, it_cut = one.solve(b, x0, 1e-9, 38)
    assert restarts[0] > 0 and it_cut == 39 < it_one      # (the reference's loop counter has passed itmax: src/solvers.f90:59)

    def body(m, r):
        m.assemble_poisson(sdx, sdy, sdz)
        x, it = m.solve(b, x0, 1e-9, 5000)
        rs = m.slab(0)[0].restart_count()
        xc, itc = m.solve(b, x0, 1e-9, 38)
        return x, it, rs, xc, itc

    got = run_ranks(E, world, body)
    assert [g[1] for g in got] == [it_one] * world and [g[2] for g in got] == restarts
    assert np.array_equal(merge(x0, [g[0] for g in got]), x_one)
    assert [g[4] for g in got] == [39] * world
    assert np.array_equal(merge(x0, [g[3] for g in got]), x_cut)


@pytest.mark.parametrize("world, pitched", [(2, False), (3, False), (2, True)], ids=["2", "3", "2-both-splits"])
def test_av_rank_job_reproduces_the_reference_capture(E, monkeypatch, world, pitched):
    """The A-V system [Ax | Ay | Az | U] (src/EC3D.f90:408) cut into slabs of the structured form: four blocks exchanged per
    neighbour, the U block two planes deep.  Two time steps of the reference's capture: its iteration counts, and the
    one-process handle's x bit for bit."""
    from conftest import load_golden
    set_knobs(monkeypatch, **(dict(SLAB_PLAN=5) if pitched else {}))
    if pitched:     # tile-aligned planes: the slabs can split K1 / K3 too: plan 5 (the exchange behind two launches; asked for)
        monkeypatch.setenv("EC3D_PITCH", "2")
    else:
        monkeypatch.delenv("EC3D_PITCH", raising=False)
    g = load_golden("g2_conducting_hole_16x15x14")
    tol, itmax = float(g["tol"]), int(g["itmax"])
    geo = (g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
    with E.EC3DMulti(world, devices=[0] * world) as one:
        one.assemble(*geo)
        want_plan = one.plan()
        assert want_plan[0] == (5 if pitched else 2)
        ref = [one.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax) for k in (0, 1)]

    def body(m, r):
        m.assemble(*geo)
        assert m.plan() == want_plan
        return [m.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax) for k in (0, 1)]

    got = run_ranks(E, world, body)
    for k in (0, 1):
        assert [got[r][k][1] for r in range(world)] == [ref[k][1]] * world
        x = merge(g[f"xin{k}"], [got[r][k][0] for r in range(world)])
        assert np.array_equal(x, ref[k][0])
        assert abs(ref[k][1] - int(g["iters"][k])) <= 1
        assert np.linalg.norm(x - g[f"xout{k}"]) <= 10 * tol * np.linalg.norm(g[f"xout{k}"])


def test_the_transport_itself(E, monkeypatch):
    """ec3d_rccl_loopback_selftest (tests/support/rccl_loopback.cpp): two ranks exchange and gather known values in the order
    of the driver's calls (values checked on the device side of each rank), and a receive that meets a send of another
    length is an error on the spot (real RCCL would hang or write past the buffer) -- so a wrong halo piece in the driver
    cannot pass the tests above silently."""
    import ctypes as C
    set_knobs(monkeypatch)
    E.load_library()                       # (one HIP runtime in the process: the product library's)
    T = C.CDLL(LOOPBACK, mode=C.RTLD_LOCAL)
    T.ec3d_rccl_loopback_selftest.restype = C.c_int
    assert T.ec3d_rccl_loopback_selftest() == 0


def test_the_product_library_holds_no_transport_double(E):
    """The loopback transport is test infrastructure: libec3d_hip.so neither contains it nor looks for a switch of its
    own -- a stand-in has to be NAMED (EC3D_RCCL_LIB=<path>), and naming one is announced on stderr."""
    import subprocess
    from eddy_currents_3d_amd import build
    syms = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "loopback" not in syms.lower()
    blob = open(build.LIB, "rb").read()
    assert b"EC3D_RCCL_LOOPBACK" not in blob and b"EC3D_RCCL_LIB" in blob


@pytest.mark.parametrize("name, moving, world", [("g2_conducting_hole_16x15x14", False, 2), ("g3_moving_coil_18x16x12", True, 3)])
def test_time_loop_state_on_rank_handles(E, monkeypatch, name, moving, world):
    """src/EC3D.f90:370-404 (RHS build) and :412-433 (post-update) on the rank handles: every rank's rows of the assembled b
    and of the post-updated x are the reference's loop state bit for bit (host vectors are global on every rank)."""
    from conftest import load_golden
    from test_gpu_timeloop import coil_sources
    set_knobs(monkeypatch)
    g = load_golden(name)
    n = len(g["irow"]) - 1
    steps = len(g["iters"])

    def body(m, r):
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        m.upload("X", np.zeros(n))
        m.upload("B", np.zeros(n))
        xs, bs = [], []
        for k in range(steps):
            if k > 0:
                m.upload("X", g[f"xout{k - 1}"])
                m.upload("B", g[f"b{k - 1}"])
                m.post_update()
                xs.append(m.download("X"))
            idx, val = coil_sources(g, k, moving)
            m.rhs_step(idx, val, moving=moving)
            bs.append(m.download("B"))
        return xs, bs

    got = run_ranks(E, world, body)
    # download fills this rank's rows of a zero vector: the rows of all ranks together are the whole vector
    for k in range(steps):
        assert np.array_equal(merge(np.zeros(n), [got[r][1][k] for r in range(world)]), g[f"b{k}"]), f"b of step {k}"
        if k > 0:
            assert np.array_equal(merge(np.zeros(n), [got[r][0][k - 1] for r in range(world)]), g[f"xin{k}"]), f"x of step {k}"


def keep(parts):
    """(the parts are views of buffers the handle owns: copied before the handle goes)"""
    return {k: (None if v is None else [np.array(a) for a in v]) for k, v in parts.items()}


def test_fields_and_csr_route_on_rank_handles(E, monkeypatch):
    """Field output (src/EC3D.f90:241-366: B = curl A, the eddy current density) and the drop-in's way in (the reference's
    CSR triple, cut by every rank for itself) on three rank handles: each rank's cells equal the matching slab of the
    one-process handle, whose file bytes tests/test_gpu_multi.py pins to the reference's."""
    from conftest import load_golden
    set_knobs(monkeypatch)
    world = 3
    g = load_golden("g3_moving_coil_18x16x12")
    tol, itmax = float(g["tol"]), int(g["itmax"])
    geo = (g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
    with E.EC3DMulti(world, devices=[0] * world) as one:
        one.assemble(*geo)
        one.upload("X", g["xout1"])
        one.upload("B", g["b1"])
        one.post_update()
        slot = one.vtk_fields_begin(g["delta"], big_endian=True)
        want = keep(one.vtk_fields_wait(slot, big_endian=True))
        x_one, it_one = one.solve(g["b0"], g["xin0"], tol, itmax)

    def body(m, r):
        m.assemble(*geo)
        m.upload("X", g["xout1"])
        m.upload("B", g["b1"])
        m.post_update()
        slot = m.vtk_fields_begin(g["delta"], big_endian=True)
        parts = keep(m.vtk_fields_wait(slot, big_endian=True))
        m.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        return parts, m.solve(g["b0"], g["xin0"], tol, itmax)

    got = run_ranks(E, world, body)
    for r in range(world):
        parts = got[r][0]
        assert set(parts) == set(want)
        for key in want:
            if parts[key] is None:     # (a rank without a conductor has no eddy part; the one-process handle fills zeros)
                assert key == "eddy" and not np.any(want[key][r])
                continue
            assert len(parts[key]) == 1 and np.array_equal(parts[key][0], want[key][r]), (r, key)
        assert got[r][1][1] == it_one
    assert np.array_equal(merge(g["xin0"], [got[r][1][0] for r in range(world)]), x_one)


def test_av_rank_job_with_uneven_u_exchange(E, monkeypatch):
    """Five ranks of an A-V job on the LIM geometry resampled to 64 x 32 x 48 (tests/golden/g4_LIM; conductor in planes 20 .. 27,
    cuts at 9, 19, 28, 38): the conductor ends next to two of the four cuts, so there a rank sends its two U planes down but
    receives none from below (or the other way round), and the outer cuts carry no U plane at all.  The k-th send to a neighbour must meet
    its k-th receive from me with the same length (the loopback transport checks every pair): x and iter of the rank job
    equal the one-process handle's bit for bit, on plan 5 (K1 / K3 split as well)."""
    from conftest import load_golden
    from eddy_currents_3d_amd import vxc
    set_knobs(monkeypatch, SLAB_PLAN=5)
    monkeypatch.setenv("EC3D_AV_SEND_EMPTY_U", "0")      # (both opt-in until two real devices have verified them)
    monkeypatch.delenv("EC3D_PITCH", raising=False)
    world = 5
    g = load_golden("g4_LIM")
    model = vxc.resample(vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                                      tuple(float(x) for x in g["adj"])), 64, 32, 48)
    t = vxc.domain_tables(model)
    geo = (t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
    with E.EC3DMulti(world, devices=[0] * world) as one:
        one.assemble(*geo)
        n = one.n
        want_plan = one.plan()
        rows = [one.halo_rows(r) for r in range(world)]
        b = one.spmv(np.random.Generator(np.random.PCG64(3)).standard_normal(n))
        x_one, it_one = one.solve(b, np.zeros(n), 1e-8, 5000)
    assert want_plan[0] == 5
    assert any(s != r for s, r in rows) and len({s for s, _ in rows}) >= 2      # uneven, and not the same on every rank

    def body(m, r):
        m.assemble(*geo)
        assert m.plan() == want_plan and m.halo_rows(0) == rows[r]
        return m.solve(b, np.zeros(n), 1e-8, 5000)

    got = run_ranks(E, world, body)
    assert [g_[1] for g_ in got] == [it_one] * world
    assert np.array_equal(merge(np.zeros(n), [g_[0] for g_ in got]), x_one)
