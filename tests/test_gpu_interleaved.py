"""The INTERLEAVED z-march of the structured A-V kernels (round 6; csrc/ec3d_kernels.hip walk_zm_il, Sweep::il_*).

On a single-rank handle with tile-aligned planes a workgroup of K1 / K3 / the bare SpMV / the residual kernel visits, plane
by plane of its column, the tiles of A_x, A_y, A_z and -- where it holds an unknown -- the U tile, instead of sweeping the
three A blocks one after the other and the U tiles from a list at the end: a row's coupling operands (src/EC3D.f90:656-711,
:766-959) are then lines the workgroup fetched a step earlier.  The library takes it from 25 Mi streamed rows (BASELINE
config 3 at 256^3); EC3D_SAV_IL=2 forces it onto the small captured systems here, where the GPU-order twin can follow:

  * every tile of the three A blocks and every listed U tile is visited exactly once, in the interleaved order;
  * A*x is bit-identical to the oracle's CSR SpMV (src/solvers.f90:54-61);
  * x, iter and the residual history of every captured call (warm starts included) are bit-identical to the twin, under
    the cache policies these sizes run with, from CSR and from the native assembly.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ["g2_conducting_hole_16x15x14", "g2v_conducting_moving_16x15x14", "g3_moving_coil_18x16x12"]


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


def interleaved(monkeypatch, nt="1", keep=None, nblk=None):
    monkeypatch.setenv("EC3D_SAV_IL", "2")
    monkeypatch.setenv("EC3D_PITCH", "2")        # tile-aligned planes also on these small grids
    monkeypatch.setenv("EC3D_SAV_PATCH", "0")
    monkeypatch.setenv("EC3D_NT", nt)
    if keep is None:
        monkeypatch.delenv("EC3D_KEEP", raising=False)
    else:
        monkeypatch.setenv("EC3D_KEEP", keep)
    if nblk is None:
        monkeypatch.delenv("EC3D_NBLK_SPMV", raising=False)
    else:
        monkeypatch.setenv("EC3D_NBLK_SPMV", str(nblk))


def check_visit_order(s):
    """Interleaved, complete, nothing twice."""
    g = s.geometry(1)
    off, tiles = s.visit_order(1)
    ul = np.asarray(s.ulist(), np.int64)
    front = int(g.ntiles_front)
    assert front % 3 == 0
    blk = front // 3
    want = np.sort(np.concatenate([np.arange(front, dtype=np.int64), ul]))
    assert np.array_equal(np.sort(tiles.astype(np.int64)), want)
    assert g.ulist_n == 0 and len(ul) > 0                      # the U tiles are inside the march, not behind it
    seen_u = 0
    for w in range(len(off) - 1):
        v = tiles[off[w]:off[w + 1]].astype(np.int64)
        i = 0
        while i < len(v):
            assert v[i] < blk and v[i + 1] == v[i] + blk and v[i + 2] == v[i] + 2 * blk
            i += 3
            if i < len(v) and v[i] >= front:
                assert v[i] == v[i - 3] + 3 * blk
                seen_u += 1
                i += 1
    assert seen_u == len(ul)


@pytest.mark.parametrize("nblk", [None, 8, 40])
@pytest.mark.parametrize("name", CASES)
def test_visit_order_and_spmv_bitwise(E, oracle, name, nblk, monkeypatch):
    interleaved(monkeypatch, nblk=nblk)
    g = load_golden(name)
    n = len(g["irow"]) - 1
    x = np.random.Generator(np.random.PCG64(11)).standard_normal(n)
    with E.EC3DSolver() as s:
        s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        assert s.info.tail_rows == 0 and s.info.dict_classes > 0        # structured form
        check_visit_order(s)
        assert np.array_equal(s.spmv(x), oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x))
        b = g["b0"]
        s.upload("B", b)
        s.upload("X", x)
        rel, bn = s.true_residual()
        r = b - oracle.spmv_csr(g["valA"], g["irow"], g["jcol"], x)
        assert bn == pytest.approx(float(np.linalg.norm(b)), rel=1e-13)
        assert rel == pytest.approx(float(np.linalg.norm(r) / np.linalg.norm(b)), rel=1e-12)


@pytest.mark.parametrize("policy", [("0", None), ("1", "0"), ("1", "11")], ids=lambda p: f"nt{p[0]}-keep{p[1]}")
@pytest.mark.parametrize("route", ["csr", "native"])
@pytest.mark.parametrize("name", CASES)
def test_solve_bitwise_vs_twin(E, oracle, name, route, policy, monkeypatch):
    interleaved(monkeypatch, nt=policy[0], keep=policy[1])
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s:
        if route == "csr":
            s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
        else:
            s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        check_visit_order(s)
        for k in range(len(g["iters"])):
            x, it, hist = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax, hist_cap=400)
            xo, ito, hs, hr = oracle.twin_solve(s, g["valA"], g["irow"], g["jcol"], g[f"b{k}"], g[f"xin{k}"], tol, itmax,
                                                hist_cap=400)
            assert it == ito
            assert np.array_equal(x, xo)
            assert np.array_equal(hist[:it, 0], hs[:it])
            if it > 1:
                assert np.array_equal(hist[:it - 1, 1], hr[:it - 1])


def test_off_below_the_threshold_and_when_switched_off(E, monkeypatch):
    """The library's own rule: the separate U list on small systems (the fixtures) and with EC3D_SAV_IL=0."""
    for il in (None, "0"):
        interleaved(monkeypatch)
        if il is None:
            monkeypatch.delenv("EC3D_SAV_IL")
        else:
            monkeypatch.setenv("EC3D_SAV_IL", il)
        g = load_golden(CASES[0])
        with E.EC3DSolver() as s:
            s.set_matrix_csr(g["valA"], g["irow"], g["jcol"])
            assert s.geometry(1).ulist_n == len(s.ulist()) > 0
