""".vxc ingest / writer / refiner, CPU only.  Pinned by the golden fixtures: their geometry
tables reproduce the unmodified reference's CSR bit for bit (tests/test_oracle_golden.py), and their
.vxc inputs are the ones the reference itself ingested (oracle/make_goldens.py)."""
import os

import numpy as np
import pytest

from conftest import load_golden
from eddy_currents_3d_amd import vxc


def golden_model(case):
    from oracle import make_goldens as G
    inp = {"g1": G.inputs_g1, "g2": G.inputs_g2, "g2v": lambda: G.inputs_g2(vel=True), "g3": G.inputs_g3}[case]()
    return vxc.VxcModel(inp["vox"], inp["names"], float(inp["lattice_dim"]), tuple(float(a) for a in inp["adj"]))


@pytest.mark.parametrize("comp", ["ZLIB", "ASCII_READABLE"])
def test_write_read_roundtrip(tmp_path, comp):
    m = golden_model("g2")
    p = str(tmp_path / "a.vxc")
    vxc.write_vxc(p, m, compression=comp)
    r = vxc.read_vxc(p)
    assert r.compression == comp and np.array_equal(r.vox, m.vox) and r.names == m.names
    assert r.lattice_dim == m.lattice_dim and r.adj == m.adj


@pytest.mark.parametrize("case,fixture", [("g1", "g1_nonconducting_8x7x6"), ("g2", "g2_conducting_hole_16x15x14"),
                                          ("g2v", "g2v_conducting_moving_16x15x14"),
                                          ("g3", "g3_moving_coil_18x16x12")])
def test_domain_tables_match_fixtures(case, fixture):
    g = load_golden(fixture)
    t = vxc.domain_tables(golden_model(case))
    assert np.array_equal(t["geoPHYS"], g["geoPHYS"])
    assert np.array_equal(t["geoPHYS_C"], g["geoPHYS_C"])
    assert np.array_equal(t["valPHYS"], g["valPHYS"])
    assert np.array_equal(t["delta"], g["delta"])
    assert t["dt"] == float(g["dt"]) and t["tol"] == float(g["tol"]) and t["itmax"] == int(g["itmax"])
    assert np.array_equal(t["BND"], g["BND"])


def test_numeric_prefixes():
    assert vxc.numeric("5m") == 5e-3 and vxc.numeric("0.4m") == 0.4e-3 and vxc.numeric("1u") == 1e-6
    assert vxc.numeric("10000") == 10000.0 and vxc.numeric("2meg") == 2e6 and vxc.numeric("1k3") == 1300.0
    assert vxc.numeric("1e-3") == 1e-3 and vxc.numeric("40m") == 0.04


def test_evaluate_constants():
    c = dict(MU0=vxc.MU0, DX=0.00333, DZ=0.00333, PI=np.pi)
    assert vxc.evaluate("'mu0*35.26e6'", c) == vxc.MU0 * 35.26e6
    assert vxc.evaluate("'183/(6*dx*6*dz)'", c) == 183 / (6 * 0.00333 * 6 * 0.00333)
    with pytest.raises(NotImplementedError):
        vxc.evaluate("'cos(2*pi)'", c)


def test_refine_keeps_physical_size():
    m = golden_model("g2")
    r = vxc.refine(m, 2, 3, 4)
    assert r.vox.shape == (14 * 4, 15 * 3, 16 * 2)
    assert np.allclose(r.delta * np.array([2, 3, 4]), m.delta)
    assert np.array_equal(r.vox[::4, ::3, ::2], m.vox)
    t = vxc.domain_tables(r)
    assert t["ncells0"] == 24 * int((m.vox == 1).sum())


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="reference inputs not present")
@pytest.mark.parametrize("stem", ["compare_to_Elmer", "ec_src_move_hole", "LIM"])
def test_shipped_inputs_ingest(stem):
    """The three shipped ZLIB files decode to the grids the reference reports and to
    the unknown counts captured from the reference run (g4 fixtures)."""
    g = load_golden("g4_" + stem)
    m = vxc.read_vxc(f"/root/reference/src/{stem}.vxc")
    assert m.compression == "ZLIB"
    assert np.array_equal(m.vox, g["vox"])
    t = vxc.domain_tables(m)
    assert 3 * m.vox.size + t["ncells0"] == int(g["n"])
    assert t["tol"] == float(g["tol"])
