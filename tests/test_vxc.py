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


def test_resample_is_refine_for_integer_factors_and_keeps_the_physical_size():
    import numpy as np
    from eddy_currents_3d_amd import vxc
    rng = np.random.Generator(np.random.PCG64(3))
    m = vxc.VxcModel(rng.integers(0, 4, (5, 6, 7)).astype(np.uint8), ["a D=1", "b D=1", "c D=1"], 0.004, (1.0, 1.25, 0.75))
    assert np.array_equal(vxc.resample(m, 14, 18, 10).vox, vxc.refine(m, 2, 3, 2).vox)
    big = vxc.resample(m, 19, 13, 11)                       # no integer factor on any axis
    assert big.vox.shape == (11, 13, 19)
    # every new voxel takes the material of the old voxel that contains its centre
    for (i, j, k) in [(0, 0, 0), (18, 12, 10), (9, 6, 5), (3, 11, 7)]:
        io, jo, ko = int((i + 0.5) * 7 / 19), int((j + 0.5) * 6 / 13), int((k + 0.5) * 5 / 11)
        assert big.vox[k, j, i] == m.vox[ko, jo, io]
    ext_old = np.array(m.delta) * np.array([7, 6, 5])
    ext_new = np.array(big.delta) * np.array([19, 13, 11])
    assert np.allclose(ext_new, ext_old, rtol=1e-7)        # to the 10 characters the reference reads


def test_lattice_numbers_follow_the_reference_10_character_buffer(tmp_path):
    """src/vxc2data.f90:50: CHARACTER(len=10) ch_e -- the reference reads the first 10 characters of Lattice_Dim and
    the *_Dim_Adj values whatever the file says; resample() only produces numbers that survive that, write_vxc
    refuses others, read_vxc cuts like the reference."""
    import numpy as np
    import pytest
    from eddy_currents_3d_amd import vxc
    m = vxc.VxcModel(np.zeros((22, 32, 176), np.uint8), ["a D=1"], 0.0025, (2.0, 3.8095, 4.3))
    big = vxc.resample(m, 384, 192, 128)
    assert big.adj == (0.91666666, 0.63491666, 0.7390625)
    assert all(len(repr(a)) <= 10 for a in big.adj)
    path = str(tmp_path / "m.vxc")
    vxc.write_vxc(path, vxc.VxcModel(np.zeros((2, 2, 2), np.uint8), ["a D=1"], 0.0025, big.adj), compression="ASCII_READABLE")
    assert vxc.read_vxc(path).adj == big.adj
    with pytest.raises(ValueError, match="10 characters"):
        vxc.write_vxc(path, vxc.VxcModel(np.zeros((2, 2, 2), np.uint8), ["a D=1"], 0.0025, (0.916666666667, 1.0, 1.0)))
    txt = open(path).read().replace("<X_Dim_Adj>0.91666666</X_Dim_Adj>", "<X_Dim_Adj>0.916666666667</X_Dim_Adj>")
    open(path, "w").write(txt)
    assert vxc.read_vxc(path).adj[0] == 0.91666666          # what the reference would have read
