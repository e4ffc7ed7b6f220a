"""CPU-only: the palette source language and the motion of the sources (eddy_currents_3d_amd/host.py) on the
reference's three shipped inputs.  At T = 0 nothing but the sources is in the right-hand side, so its norm is
the ||b|| the unmodified reference handed to its solver in step 0 (tests/golden/g4_*)."""
import math

import numpy as np
import pytest

from conftest import load_golden


def _program(case):
    from eddy_currents_3d_amd import host, vxc
    g = load_golden("g4_" + case)
    model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    t = vxc.domain_tables(model)
    return g, model, t, host.SourceProgram(model, t)


@pytest.mark.parametrize("case,nfun,moving", [("compare_to_Elmer", 4, False), ("ec_src_move_hole", 4, True),
                                              ("LIM", 12, True)])
def test_step0_sources_equal_the_reference_rhs(case, nfun, moving):
    g, model, t, prog = _program(case)
    assert len(prog.funs) == nfun and prog.moving == moving
    idx, val, mv = prog.step(0.0)
    assert mv == moving and idx.min() >= 1 and idx.max() <= 2 * model.vox.size   # X and Y components only
    b = np.zeros(3 * model.vox.size)
    b[idx - 1] = val
    assert np.linalg.norm(b) == pytest.approx(float(g["bnorm"][0]), rel=1e-14)


def test_source_motion_follows_the_velocity_functions():
    """ec_src_move_hole: Vsx = Vmx(t) = a*2pi*f*sin(2pi f t), Vsy = Vmy(t) = -b*2pi*f*cos(2pi f t): the coil
    starts moving in -y; the accumulated distance is rounded to whole cells (src/EC3D.f90:1052-1062)."""
    g, model, t, prog = _program("ec_src_move_hole")
    sdz, sdy, sdx = model.vox.shape
    dist = np.zeros(2)
    for k in range(5):
        T = k * t["dt"]
        idx, _, _ = prog.step(T)
        w = 2 * math.pi * 25
        dist += np.array([t["delta"][0] * (sdx - 42) / 2 * w * math.sin(w * T),
                          -t["delta"][1] * (sdy - 42) / 2 * w * math.cos(w * T)]) * t["dt"] / np.asarray(t["delta"])[:2]
        f = prog.funs[0]
        assert list(f["length"][:2]) == [int(math.floor(abs(d) + 0.5)) * (1 if d >= 0 else -1) for d in dist]
        cells0 = f["nodes"] - 1
        moved = idx[:len(cells0)] - 1
        assert np.array_equal(moved % sdx, cells0 % sdx + f["length"][0])
        assert np.array_equal((moved // sdx) % sdy, (cells0 // sdx) % sdy + f["length"][1])


def test_expression_functions():
    from eddy_currents_3d_amd.host import Expression
    v = dict(A=2.0, T=0.25)
    assert Expression("A*COSD(360*T)+IMPL2(-1)+IMPLS(0)+POS(-3)+NINT(2.5)+INT(-1.7)")(v) == pytest.approx(
        2.0 * math.cos(math.pi / 2) - 1.0 + 0.0 + 0.0 + 3.0 - 1.0)
    assert Expression("A^3-LG(100)+LN(EXP(1))+TH(0)+ATG(0)")(v) == pytest.approx(8.0 - 2.0 + 1.0)
    with pytest.raises(ValueError):
        Expression("__IMPORT__(1)")(v)


def test_source_cell_file_is_the_reference_file():
    """src_N.vtk of the moving-coil case (constant Vsx, FUNC Vsy): bytes equal to what the unmodified reference
    wrote (tests/golden/g3_src_vtk.npz, oracle/make_goldens.py case_g3_src_vtk) at every output point of a
    24-step run -- long enough for the coil to reach the clamp two cells off the box (src/EC3D.f90:1064-1114),
    where the constant-velocity motion stops through ``movestop``."""
    from eddy_currents_3d_amd import host, vxc
    from eddy_currents_3d_amd.vtk import src_vtk_bytes
    g = load_golden("g3_src_vtk")
    model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    t = vxc.domain_tables(model)
    prog = host.SourceProgram(model, t)
    sdz, sdy, sdx = model.vox.shape
    npoints = sum(1 for k in g.files if k.startswith("vtk_src_"))
    assert npoints >= 20
    T, clamped = 0.0, False
    for k in range(npoints + 1):              # output point N is written at the end of step N (N >= 1)
        prog.step(T)
        T = T + t["dt"]
        if k >= 1:
            assert src_vtk_bytes(sdx, sdy, sdz, t["delta"], prog.groups) == g[f"vtk_src_{k}"].tobytes(), k
        cells = np.concatenate([gr[1] for gr in prog.groups]) - 1
        clamped |= bool(((cells % sdx + 1) >= sdx - 2).any())
    assert clamped, "the run was meant to reach the clamp"


@pytest.mark.parametrize("case", ["ec_src_move_hole", "LIM"])
def test_sources_evaluated_ahead_are_the_sequential_ones(case):
    """host._SourcesAhead (the source program of step k + 1 evaluated on a thread while step k solves; sources are
    functions of time alone, src/EC3D.f90:245-340) hands the time loop exactly what the sequential evaluation gives:
    the same (unknown id, value) lists, the same cell groups for src_N.vtk, in the same order, for the same T sequence
    (T = T + DT accumulated in floating point as the loop does), clamp and movestop state included."""
    from eddy_currents_3d_amd import host
    _, _, t, seq = _program(case)
    _, _, _, par = _program(case)
    DT, Time, steps = float(t["dt"]), float(t["time"]), 30
    ahead = host._SourcesAhead(par, 0.0, DT, Time, steps)
    T = 0.0
    for k in range(steps):
        idx, val, moving = seq.step(T)
        aidx, aval, amoving, agroups = ahead.next(T)
        assert np.array_equal(idx, aidx) and np.array_equal(val, aval) and moving == amoving
        assert len(agroups) == len(seq.groups)
        for (ax, cells, v), (bx, bcells, bv) in zip(seq.groups, agroups):
            assert ax == bx and v == bv and np.array_equal(cells, bcells)
        T = T + DT
    ahead.thread.join(timeout=10)
    assert not ahead.thread.is_alive()
