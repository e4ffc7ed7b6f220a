"""Multi-GPU inside the library (include/ec3d_hip.h section 2c, csrc/ec3d_multi.hip): one process, N slabs, one
host thread per slab, halo planes pulled over peer access, partial sums read in place.  On the one-GPU test
box every slab sits on device 0 -- the same code path with local copies instead of xGMI ones.

Bar: bit-identical to the staged drivers of eddy_currents_3d_amd/dist.py on the same slabs (same kernels, same
rank-ordered sums), which in turn are pinned against the reference's captured solutions; plus the reference
fixtures directly (src/solvers.f90:3-50 results within 10*tol, src/EC3D.f90:370-433 loop state bit for bit,
field_N.vtk bytes)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    return E


def _inprocess_poisson(sdx, sdy, sdz, world, b, tol, itmax):
    from eddy_currents_3d_amd.dist import HipSlabOps, InProcessSlabs, slab_bounds
    ops = []
    for r in range(world):
        k0, k1 = slab_bounds(sdz, r, world)
        o = HipSlabOps(sdx, sdy, sdz, k0, k1, world)
        o.set_vector("B", b.reshape(sdz, sdx * sdy)[k0:k1].reshape(-1))
        ops.append(o)
    drv = InProcessSlabs(ops)
    it = drv.solve(tol, itmax)
    x = drv.x()
    for o in ops:
        o.close()
    return x, it


@pytest.mark.parametrize("grid,world", [((64, 64, 40), 2), ((64, 64, 48), 3), ((64, 64, 40), 4),   # z-marching, overlap plan
                                        ((24, 24, 24), 2), ((20, 20, 20), 3)])                      # plain plan
def test_multi_poisson_bitwise_equals_staged_slabs(E, oracle, grid, world):
    sdx, sdy, sdz = grid
    tol = 1e-8
    b = np.random.Generator(np.random.PCG64(5)).standard_normal(sdx * sdy * sdz)
    x_ref, it_ref = _inprocess_poisson(sdx, sdy, sdz, world, b, tol, 5000)
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        assert m.n == sdx * sdy * sdz
        if sdx * sdy % 512 == 0:
            assert all(m.slab(r)[0].can_overlap() for r in range(world))
        x, it = m.solve(b, np.zeros(m.n), tol, 5000)
        # a second solve on the same handle (warm start from the solution: converges at once or nearly)
        x2, it2 = m.solve(b, x, tol, 5000)
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    res = np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b)
    print(f"{world} slabs of {sdx}x{sdy}x{sdz} in the library: iter {it} (staged driver {it_ref}), true residual {res:.2e}")
    assert it == it_ref and np.array_equal(x, x_ref)
    assert res < 5 * tol
    assert it2 <= 2


def test_multi_with_one_rank_equals_plain_handle(E, oracle):
    N, tol = 24, 1e-8
    b = oracle.bar_rhs(N)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x_ref, it_ref, _ = s.solve(b, np.zeros(N ** 3), tol, 10000)
    with E.EC3DMulti(1) as m:
        m.assemble_poisson(N, N, N)
        x, it = m.solve(b, np.zeros(N ** 3), tol, 10000)
    assert it == it_ref and np.array_equal(x, x_ref)


@pytest.mark.parametrize("name,world", [("g2_conducting_hole_16x15x14", 2), ("g2_conducting_hole_16x15x14", 3),
                                        ("g3_moving_coil_18x16x12", 2), ("g2v_conducting_moving_16x15x14", 4)])
@pytest.mark.parametrize("structured", [True, False])
def test_multi_av_slabs_match_staged_driver_and_reference(E, name, world, structured, plane_pitch, monkeypatch):
    """The full A-V system cut through the conductor; native assembly on slabs inside the library.  The staged driver runs
    the producer-side schedule (K2 / K5 boundary tiles first), so the library is held to plan 2 here; what it picks by
    itself (plan 5 where the slabs can split K1 / K3 too) is the next test's."""
    from eddy_currents_3d_amd.dist import HipAVSlabOps, InProcessSlabs, slab_bounds
    monkeypatch.setenv("EC3D_SLAB_PLAN", "2")
    g = load_golden(name)
    sdz = g["geoPHYS"].shape[0]
    n = len(g["irow"]) - 1
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DMulti(world, devices=[0] * world, structured=structured) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        assert m.n == n
        for k in (0, 1):
            ops = []
            for r in range(world):
                k0, k1 = slab_bounds(sdz, r, world)
                o = HipAVSlabOps(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]),
                                 k0, k1, world, structured=structured)
                o.set_vector_global("B", g[f"b{k}"])
                o.set_vector_global("X", g[f"xin{k}"])
                ops.append(o)
            drv = InProcessSlabs(ops, vsplit=True)
            it_ref = drv.solve(tol, itmax)
            x_ref = drv.x(n)
            for o in ops:
                o.close()
            x, it = m.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            xr = g[f"xout{k}"]
            rel = np.linalg.norm(x - xr) / np.linalg.norm(xr)
            print(f"{name} in {world} slabs (library), step {k}: iter {it} / staged {it_ref} / reference "
                  f"{int(g['iters'][k])}, rel diff vs reference {rel:.2e}")
            assert it == it_ref and np.array_equal(x, x_ref)
            assert rel <= 10 * tol
            assert abs(it - int(g["iters"][k])) <= max(3, 0.15 * int(g["iters"][k]))


@pytest.mark.parametrize("name,world", [("g2_conducting_hole_16x15x14", 2), ("g3_moving_coil_18x16x12", 2)])
def test_multi_av_slabs_both_splits_reproduce_the_reference(E, name, world, monkeypatch):
    """Plan 5 on A-V slabs of the structured form (EC3D_SLAB_PLAN=5; where every slab can split K1 / K3 as well: the
    exchange behind two launches): K1 / K3 as an interior launch -- the z-march over the window narrowed by two planes at
    both ends of every A block, then the U tiles of those planes -- and a boundary launch over ONE list of the outer
    planes' tiles of all four blocks.  Every owned tile is visited by exactly one of the two launches (visit orders 3 / 4
    against 1); AP = A P of the first iteration is the same bits as on plan 2 (the rows do not care which launch computes
    them); the reference's captured time steps come out with the reference's iteration counts and x to rounding."""
    monkeypatch.delenv("EC3D_SLAB_PLAN", raising=False)
    monkeypatch.setenv("EC3D_PITCH", "2")       # tile-aligned planes also on these small grids (the default from 4.5 Mi rows)
    g = load_golden(name)
    n = len(g["irow"]) - 1
    tol, itmax = float(g["tol"]), int(g["itmax"])
    geo = (g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
    ap = {}
    for plan in (2, 5):
        monkeypatch.setenv("EC3D_SLAB_PLAN", str(plan))
        with E.EC3DMulti(world, devices=[0] * world, structured=True) as m:
            m.assemble(*geo)
            m.upload("B", g["b0"])
            m.upload("X", g["xin0"])
            m.iterate_begin()
            m.iterate(1, 1)
            m.synchronize()
            ap[plan] = m.download("AP")
    assert np.array_equal(ap[2], ap[5]) and np.linalg.norm(ap[2]) > 0
    # plan 5 is opt-in until two real devices have verified it (csrc/ec3d_multi.hip finish_setup): the default is plan 2
    monkeypatch.delenv("EC3D_SLAB_PLAN")
    with E.EC3DMulti(world, devices=[0] * world, structured=True) as m:
        m.assemble(*geo)
        assert m.plan()[0] == 2
    monkeypatch.setenv("EC3D_SLAB_PLAN", "5")
    with E.EC3DMulti(world, devices=[0] * world, structured=True) as m:
        m.assemble(*geo)
        assert m.plan()[0] == 5
        for r in range(world):
            v = m.slab(r)[0]
            assert v.can_overlap()
            whole = np.sort(v.visit_order(1)[1])
            parts = np.sort(np.concatenate([v.visit_order(3)[1], v.visit_order(4)[1]]))
            assert np.array_equal(whole, parts) and len(np.unique(whole)) == len(whole)
        for k in range(len(g["iters"])):
            x, it = m.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            xr = g[f"xout{k}"]
            rel = np.linalg.norm(x - xr) / np.linalg.norm(xr)
            print(f"{name} in {world} slabs, plan 5, step {k}: iter {it} / reference {int(g['iters'][k])}, rel diff {rel:.2e}")
            assert it == int(g["iters"][k]) and rel <= 1e-12


@pytest.mark.parametrize("name,world", [("g2_conducting_hole_16x15x14", 2), ("g3_moving_coil_18x16x12", 3)])
def test_multi_csr_route_equals_native_slabs(E, name, world, plane_pitch):
    """What the drop-in receives (the reference's CSR triple) cut into slabs == the natively assembled slabs."""
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DMulti(world, devices=[0] * world) as a, E.EC3DMulti(world, devices=[0] * world) as c:
        a.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        c.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        assert a.n == c.n
        for k in (0, 1):
            xa, ita = a.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            xc, itc = c.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            assert ita == itc and np.array_equal(xa, xc)


@pytest.mark.parametrize("name,world", [("g2_conducting_hole_16x15x14", 2), ("g2v_conducting_moving_16x15x14", 3),
                                        ("g3_moving_coil_18x16x12", 4)])
@pytest.mark.parametrize("route", ["native", "native bands+tail", "csr"])
def test_multi_spmv_bitwise_equals_the_reference_operator(E, oracle, name, world, route, plane_pitch):
    """src/solvers.f90:54-61 over the slabs -- every slab's operator (native assembly in both storages, or cut out
    of the reference's CSR triple) and the halo exchange together: A*x equals the oracle's CSR SpMV on the
    reference's captured matrix bit for bit, cuts through the conductor included."""
    g = load_golden(name)
    n = len(g["irow"]) - 1
    x = np.random.Generator(np.random.PCG64(17)).standard_normal(n)
    want = oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x)
    with E.EC3DMulti(world, devices=[0] * world, structured=(route != "native bands+tail")) as m:
        if route == "csr":
            m.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        else:
            m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        assert np.array_equal(m.spmv(x), want)


def test_multi_spmv_of_the_cube_bitwise(E, oracle):
    sdx, sdy, sdz = 64, 64, 40
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    x = np.random.Generator(np.random.PCG64(18)).standard_normal(sdx * sdy * sdz)
    with E.EC3DMulti(4, devices=[0] * 4) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        assert np.array_equal(m.spmv(x), oracle.spmv_csr(valA, irow, jcol, x))


@pytest.mark.parametrize("name,moving,world", [("g2_conducting_hole_16x15x14", False, 2),
                                               ("g3_moving_coil_18x16x12", True, 3)])
@pytest.mark.parametrize("structured", [True, False])
def test_multi_time_loop_state_bitwise(E, name, moving, world, structured, plane_pitch):
    """src/EC3D.f90:370-404 and :412-433 on the slabs of the multi handle: assembled b and post-updated x equal
    the reference's loop state bit for bit."""
    from test_gpu_timeloop import coil_sources
    g = load_golden(name)
    n = len(g["irow"]) - 1
    with E.EC3DMulti(world, devices=[0] * world, structured=structured) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        m.upload("X", np.zeros(n))
        m.upload("B", np.zeros(n))
        for k in range(len(g["iters"])):
            if k > 0:
                m.upload("X", g[f"xout{k - 1}"])
                m.upload("B", g[f"b{k - 1}"])
                m.post_update()
                assert np.array_equal(m.download("X"), g[f"xin{k}"])
            idx, val = coil_sources(g, k, moving)
            m.rhs_step(idx, val, moving=moving)
            assert np.array_equal(m.download("B"), g[f"b{k}"]), f"step {k}"


@pytest.mark.parametrize("world", [2, 3])
def test_multi_fields_reproduce_reference_file(E, world, plane_pitch):
    from eddy_currents_3d_amd.vtk import field_vtk_bytes
    g = load_golden("g3_moving_coil_18x16x12")
    sdz, sdy, sdx = g["geoPHYS"].shape
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        for k in (1, 2):
            m.upload("X", g[f"xout{k}"])
            m.upload("B", g[f"b{k}"])
            m.post_update()
            f = m.vtk_fields(g["delta"], sdx * sdy * sdz, True)
            assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], f) == g[f"vtk_field_{k}"].tobytes()
            # the same through the overlapped entry points: per-slab parts in the file's byte order
            slot = m.vtk_fields_begin(g["delta"], big_endian=True)
            parts = m.vtk_fields_wait(slot, big_endian=True)
            assert all(len(v) == world and v[0].dtype == np.dtype(">f4") for v in parts.values())
            assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], parts) == g[f"vtk_field_{k}"].tobytes()


@pytest.mark.parametrize("dictionary", [True, False])
@pytest.mark.parametrize("dims,world", [((16, 16, 24), 2), ((16, 16, 24), 3), ((20, 12, 10), 4), ((64, 64, 40), 2),
                                        # 24 planes read as three blocks of 8: fewer than two planes per rank for the
                                        # A-V reading (status 2), yet 24 planes cut six or eight ways as a cube
                                        ((16, 16, 24), 6), ((16, 16, 24), 8)])
def test_multi_csr_route_cuts_a_single_component_cube(E, oracle, dims, world, dictionary):
    """BASELINE configs 2 and 4 as the drop-in symbol receives them: the CSR triple of the single-component 7-point
    operator (src/EC3D.f90:528-654 with no conducting cell).  A plane count that is a multiple of 3 passes the A-V
    recogniser as "three blocks" whose faces couple -- the multi route then, and whenever the A-V recogniser says
    no, reads the matrix as seven bands on a grid and cuts it plane by plane: A*x bit-identical to the oracle's CSR
    row sums, the solve bit-identical to the natively assembled slabs (same coefficients, same slabs)."""
    sdx, sdy, sdz = dims
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    n = sdx * sdy * sdz
    x = np.random.Generator(np.random.PCG64(77)).standard_normal(n)
    b = np.random.Generator(np.random.PCG64(78)).standard_normal(n)
    assert E.probe_csr_multi(valA, irow, jcol, world)[0]
    with E.EC3DMulti(world, devices=[0] * world, dictionary=dictionary) as m:
        m.set_matrix_csr(valA, irow, jcol)
        assert m.n == n
        y = m.spmv(x)
        xs, its = m.solve(b, np.zeros(n), 1e-9, 2000)
    assert np.array_equal(y, oracle.spmv_csr(valA, irow, jcol, x))
    with E.EC3DMulti(world, devices=[0] * world, dictionary=dictionary) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        xa, ita = m.solve(b, np.zeros(n), 1e-9, 2000)
    assert its == ita and np.array_equal(xs, xa)
    r = b - oracle.spmv_csr(valA, irow, jcol, xs)
    assert np.linalg.norm(r) <= 1e-7 * np.linalg.norm(b)


def test_multi_csr_route_refuses_a_matrix_without_a_grid(E, oracle):
    n = 64
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        with pytest.raises(E.EC3DError, match="not recognised"):
            m.set_matrix_csr(np.ones(n), np.arange(1, n + 2, dtype=np.int32), np.arange(1, n + 1, dtype=np.int32))
    valA, irow, jcol = oracle.poisson_csr(8, 8, 3)
    with E.EC3DMulti(4, devices=[0] * 4) as m:
        with pytest.raises(E.EC3DError, match="fewer z-planes than ranks"):
            m.set_matrix_csr(valA, irow, jcol)


def test_multi_refuses_what_it_cannot_cut(E):
    """Edge cases of the decomposition: no matrix yet, fewer planes than ranks, A-V slabs thinner than the two
    halo planes the one-sided A-U stencils need (src/EC3D.f90:697-706), a device ordinal that does not exist."""
    g = load_golden("g2_conducting_hole_16x15x14")          # 14 planes
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        with pytest.raises(E.EC3DError, match="no matrix"):
            m.upload("B", np.zeros(8))
        with pytest.raises(E.EC3DError, match="no matrix"):
            m.solve_resident(1e-3, 10)
    with E.EC3DMulti(4, devices=[0] * 4) as m:
        with pytest.raises(E.EC3DError, match="fewer z-planes than ranks"):
            m.assemble_poisson(8, 8, 3)
    with E.EC3DMulti(8, devices=[0] * 8) as m:
        with pytest.raises(E.EC3DError, match="at least two z-planes"):
            m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
    with pytest.raises(E.EC3DError, match="out of range"):
        E.EC3DMulti(2, devices=[0, 99])
    with pytest.raises(ValueError):
        E.EC3DMulti(2, devices=[0])


def test_multi_needs_the_devices_it_is_asked_for(E):
    import torch
    have = torch.cuda.device_count()
    with pytest.raises(E.EC3DError, match=f"needs {have + 1} devices"):
        E.EC3DMulti(have + 1)


def test_multi_bench_steps_run_and_time(E):
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        m.assemble_poisson(64, 64, 40)
        m.upload("B", np.random.Generator(np.random.PCG64(1)).standard_normal(m.n))
        m.upload("X", np.zeros(m.n))
        m.iterate_begin()
        m.iterate(1, 5)
        m.synchronize()
        ms = m.iterate(6, 5, per_kernel=True)
    assert set(ms) == {"k1", "k2", "k3", "k4", "k5"} and all(v > 0 for v in ms.values())


def test_dropin_symbol_uses_several_gpus_when_the_environment_says_so():
    """sprsbcgstabwr_ (src/solvers.f90:3) with EC3D_NGPU=2: same call, same answer as one GPU to the solver
    tolerance, the reference's iteration count within the slab bound; a child process because the switch is
    read once."""
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
import eddy_currents_3d_amd as E
g = np.load(%r)
tol, itmax = float(g["tol"]), int(g["itmax"])
out = []
for k in (0, 1):
    x = g[f"xin{k}"].copy()
    it = E.sprsBCGstabWR(g["valA"], g["irow"], g["jcol"], len(g["irow"]) - 1, g[f"b{k}"], x, tol, itmax)
    xr = g[f"xout{k}"]
    out.append((it, float(np.linalg.norm(x - xr) / np.linalg.norm(xr))))
print("RESULT", out)
""" % (REPO, os.path.join(REPO, "tests", "golden", "g2_conducting_hole_16x15x14.npz"))
    g = load_golden("g2_conducting_hole_16x15x14")
    res = {}
    for ngpu in (1, 2):
        env = dict(os.environ, EC3D_NGPU=str(ngpu), EC3D_DEVICES="0,0" if ngpu == 2 else "0")
        if ngpu == 1:
            env.pop("EC3D_DEVICES")
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        res[ngpu] = eval([l for l in p.stdout.splitlines() if l.startswith("RESULT")][0][7:])
    print(res)
    tol = float(g["tol"])
    for ngpu in (1, 2):
        for k, (it, rel) in enumerate(res[ngpu]):
            assert rel <= 10 * tol
            assert abs(it - int(g["iters"][k])) <= max(3, 0.15 * int(g["iters"][k]))


def test_dropin_symbol_cuts_a_cube_csr_over_two_gpus(oracle, tmp_path):
    """BASELINE config 2's matrix handed to sprsbcgstabwr_ (src/solvers.f90:3) as CSR with EC3D_NGPU=2: the library
    cuts it plane by plane (stderr must not say the switch was ignored) and returns the one-GPU answer to the solver
    tolerance."""
    sdx, sdy, sdz = 24, 20, 18
    valA, irow, jcol = oracle.poisson_csr(sdx, sdy, sdz)
    n = sdx * sdy * sdz
    b = np.random.Generator(np.random.PCG64(5)).standard_normal(n)
    f = str(tmp_path / "cube.npz")
    np.savez(f, valA=valA, irow=irow, jcol=jcol, b=b)
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
import eddy_currents_3d_amd as E
g = np.load(%r)
x = np.zeros(len(g["b"]))
it = E.sprsBCGstabWR(g["valA"], g["irow"], g["jcol"], len(g["irow"]) - 1, g["b"], x, 1e-10, 5000)
np.save(sys.argv[1], x)
print("RESULT", it)
""" % (REPO, f)
    xs = {}
    for ngpu in (1, 2):
        env = dict(os.environ, EC3D_NGPU=str(ngpu))
        env.pop("EC3D_DEVICES", None)
        if ngpu == 2:
            env["EC3D_DEVICES"] = "0,0"
        out = str(tmp_path / f"x{ngpu}.npy")
        p = subprocess.run([sys.executable, "-c", code, out], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "ignored" not in p.stderr, p.stderr
        xs[ngpu] = np.load(out)
    for x in xs.values():
        assert np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) <= 1e-8 * np.linalg.norm(b)
    assert np.linalg.norm(xs[2] - xs[1]) <= 1e-7 * np.linalg.norm(xs[1])


def test_teardown_with_torch_objects_gone_first():
    """Pin of the round-1 teardown hang: the library must never touch an adopted torch stream that its owner has
    already destroyed.  Destroy the torch side first, then the handle; must return (pytest-timeout guards)."""
    import gc
    import torch
    from eddy_currents_3d_amd.dist import HipSlabOps
    o = HipSlabOps(64, 64, 32, 0, 16, 2)
    o.set_vector("B", np.ones(o.n))
    o.step(0, 0, 1e-8)
    o.synchronize()
    local = o.local
    o.local = None                  # keep close() from detaching politely
    del o.store, o.lsum, o.gsum
    o.stream = None
    del o
    gc.collect()
    torch.cuda.empty_cache()
    local.close()                   # ec3d_destroy with the adopted stream and vectors already gone


def test_single_rank_entry_points_refuse_a_dist_configured_handle(E):
    """a full-grid handle configured with nranks = 1 takes its sums from gsum, which only the staged
    driver fills -- ec3d_solve / ec3d_iterate / ec3d_time_* must refuse it, and leaving dist mode restores them."""
    import torch
    N = 16
    b = np.ones(N ** 3)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x_ref, it_ref, _ = s.solve(b, np.zeros(N ** 3), 1e-8, 1000)
        lsum = torch.zeros(8, dtype=torch.float64, device="cuda")
        gsum = torch.zeros(8, dtype=torch.float64, device="cuda")
        s.dist_configure(1, lsum.data_ptr(), gsum.data_ptr())
        with pytest.raises(E.EC3DError, match="multi-rank"):
            s.solve(b, np.zeros(N ** 3), 1e-8, 1000)
        with pytest.raises(E.EC3DError, match="multi-rank"):
            s.time_iterations(2)
        with pytest.raises(E.EC3DError, match="multi-rank"):
            s.iterate(1, 1)
        s.dist_configure(1, 0, 0)
        x, it, _ = s.solve(b, np.zeros(N ** 3), 1e-8, 1000)
        assert it == it_ref and np.array_equal(x, x_ref)


# ---- a machine with at least two GPUs (skipped on the one-GPU test box).  Both run in CHILD processes with their
# own time limits: nobody has run them on two devices yet, and a deadlock there must fail one test, not end the run
# (the library's watchdog aborts a process whose threads are stuck inside the runtime).
def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2


_TWO_DEVICE_SCRIPT = r"""
import sys
import numpy as np
sys.path.insert(0, %(repo)r)
import eddy_currents_3d_amd as E
sdx, sdy, sdz, tol = 64, 64, 48, 1e-8
b = np.random.Generator(np.random.PCG64(5)).standard_normal(sdx * sdy * sdz)
res, av = {}, {}
g = np.load(%(golden)r)
for devs in ([0, 0], [0, 1]):
    with E.EC3DMulti(2, devices=devs) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        res[tuple(devs)] = m.solve(b, np.zeros(m.n), tol, 5000)
    with E.EC3DMulti(2, devices=devs) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        av[tuple(devs)] = m.solve(g["b0"], g["xin0"], float(g["tol"]), int(g["itmax"]))
for r in (res, av):
    assert r[(0, 0)][1] == r[(0, 1)][1] and np.array_equal(r[(0, 0)][0], r[(0, 1)][0]), "two devices differ from one"
print("TWO_DEVICES_OK", res[(0, 1)][1], av[(0, 1)][1])
"""


def test_multi_on_two_real_devices_equals_two_slabs_on_one():
    """Peer copies over xGMI, remote reads of the partial sums and cross-device event waits must give exactly what
    the same two slabs give on one card."""
    if not _two_devices():
        pytest.skip("needs 2 GPUs")
    code = _TWO_DEVICE_SCRIPT % dict(repo=REPO, golden=os.path.join(REPO, "tests", "golden",
                                                                    "g2_conducting_hole_16x15x14.npz"))
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, EC3D_MULTI_WATCHDOG="30"),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "TWO_DEVICES_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


_NCCL_SCRIPT = r"""
import os, sys, datetime
import numpy as np
sys.path.insert(0, %(repo)r)
import torch
import torch.distributed as dist
from eddy_currents_3d_amd.dist import SlabSolver
from bench import bar_rhs
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120),
                        device_id=torch.device("cuda", rank))
try:
    s = SlabSolver.poisson_cube(64, rank, world, device=rank)
    s.set_rhs(bar_rhs(64, s.k0, s.k1), np.zeros(s.n_local))
    it = s.solve(1e-8, 20000)
    x = s.gather_x()
    if rank == 0:
        np.save(%(out)r, np.concatenate([[it], x]))
    s.ops.close()
finally:
    dist.destroy_process_group()
"""


def test_two_ranks_over_rccl_match_the_undivided_solve(E, tmp_path):
    """one 2-rank nccl test -- in-place P2P on the ghost planes, all_gather on the adopted stream, the
    halo_start / halo_wait overlap of dist.py -- against the single-GPU solve."""
    if not _two_devices():
        pytest.skip("needs 2 GPUs")
    import socket
    from bench import bar_rhs
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "x.npy")
    script = str(tmp_path / "rank.py")
    open(script, "w").write(_NCCL_SCRIPT % dict(repo=REPO, out=out))
    procs = [subprocess.Popen([sys.executable, script], env=dict(os.environ, RANK=str(r), WORLD_SIZE="2",
                                                                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        pytest.fail("two ranks over RCCL did not finish within 300 s")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r = np.load(out)
    it, x = int(r[0]), r[1:]
    b = bar_rhs(64)
    with E.EC3DSolver() as s:
        s.assemble_poisson(64, 64, 64)
        xr, itr, _ = s.solve(b, np.zeros(64 ** 3), 1e-8, 20000)
        res = np.linalg.norm(b - s.spmv(x)) / np.linalg.norm(b)
    print(f"2 ranks over RCCL: iter {it} / undivided {itr}, true residual {res:.2e}")
    assert res < 5e-8
    assert np.linalg.norm(x - xr) <= 1e-5 * np.linalg.norm(xr)


@pytest.mark.parametrize("world", [2, 4])
def test_av_slabs_on_a_resampled_model_cover_and_skip_exactly(E, world, monkeypatch):
    """The LIM geometry (tests/golden/g4_LIM: the shipped input's voxels) resampled to 64 x 32 x 48: the conductor lies
    between two cuts, so most cuts pass through air.  (a) K1 / K3 of plan 5 as interior + boundary launch visit every owned
    tile exactly once on every slab; (b) with EC3D_AV_SEND_EMPTY_U=0 a cut's two U planes that hold no conductor cell stay at home -- the
    solve is the SAME BITS as with EC3D_AV_SEND_EMPTY_U=1, which sends them (their rows are zero in every vector), and the
    slabs really exchange less; (c) the solution solves the system (true residual, computed on the device)."""
    from eddy_currents_3d_amd import vxc
    g = load_golden("g4_LIM")
    model = vxc.resample(vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                                      tuple(float(x) for x in g["adj"])), 64, 32, 48)
    t = vxc.domain_tables(model)
    geo = (t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
    monkeypatch.setenv("EC3D_SLAB_PLAN", "5")
    out = {}
    for send_all in ("1", "0"):
        monkeypatch.setenv("EC3D_AV_SEND_EMPTY_U", send_all)
        with E.EC3DMulti(world, devices=[0] * world) as m:
            m.assemble(*geo)
            assert m.plan()[0] == 5
            n = m.n
            b = m.spmv(np.random.Generator(np.random.PCG64(3)).standard_normal(n))       # a right-hand side in the range of A
            for r in range(world):
                v = m.slab(r)[0]
                whole = np.sort(v.visit_order(1)[1])
                parts = np.sort(np.concatenate([v.visit_order(3)[1], v.visit_order(4)[1]]))
                assert v.can_overlap() and np.array_equal(whole, parts) and len(np.unique(whole)) == len(whole)
            x, it = m.solve(b, np.zeros(n), 1e-8, 5000)
            rel, _ = m.true_residual()
            out[send_all] = (x, it, [m.halo_rows(r) for r in range(world)])
            assert it <= 5000 and rel < 5e-8
    assert out["0"][1] == out["1"][1] and np.array_equal(out["0"][0], out["1"][0])
    sent = {k: sum(a for a, _ in v[2]) for k, v in out.items()}
    recv = {k: sum(b for _, b in v[2]) for k, v in out.items()}
    print(f"{world} slabs: rows sent per exchange {sent['1']} with the empty U planes, {sent['0']} without")
    assert sent["0"] == recv["0"] and sent["1"] == recv["1"]
    # two slabs: the one cut passes through the conductor, everything travels; four: two of the three cuts lie in air
    assert sent["0"] == sent["1"] if world == 2 else sent["0"] < sent["1"]
