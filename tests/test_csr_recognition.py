"""CPU-only: the host-side recogniser that decides how a CSR matrix is stored on the device
(eddy_currents_3d_amd/csrc/ec3d_sav_csr.cpp via ec3d_probe_csr).  The reference's own matrices
(tests/golden, assembled by the unmodified /root/reference/src/EC3D.f90:465-1049) must be taken into the
structured A-V form; anything whose rows would be summed in another order, or that has an entry off the
stencil, must not."""
import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def E():
    from eddy_currents_3d_amd import build
    build.build()
    import eddy_currents_3d_amd as E
    return E


@pytest.mark.parametrize("name", ["g1_nonconducting_8x7x6", "g2_conducting_hole_16x15x14",
                                  "g2v_conducting_moving_16x15x14", "g3_moving_coil_18x16x12"])
def test_reference_matrices_are_recognised(E, name):
    g = load_golden(name)
    sdz, sdy, sdx = g["geoPHYS"].shape
    p = E.probe_csr(g["valA"], g["irow"], g["jcol"])
    assert p.structured == 1
    assert (p.sdx, p.sdy, p.sdz) == (sdx, sdy, sdz)
    assert p.n_cond == int(np.count_nonzero(g["geoPHYS_C"]))
    assert p.n_cond == len(g["irow"]) - 1 - 3 * sdx * sdy * sdz
    assert 1 < p.classes <= 256
    assert p.plane_pitch == sdx * sdy        # small grids: planes are not padded


def test_plane_pitch_forced(E, monkeypatch):
    g = load_golden("g2_conducting_hole_16x15x14")
    monkeypatch.setenv("EC3D_PITCH", "2")
    p = E.probe_csr(g["valA"], g["irow"], g["jcol"])
    assert p.structured == 1 and p.plane_pitch == 512


def _swap(a, i, j):
    a[[i, j]] = a[[j, i]]


def test_rows_in_another_stored_order_are_rejected(E):
    g = load_golden("g2_conducting_hole_16x15x14")
    for r in (0, len(g["irow"]) - 2):        # an A row, a U row
        valA, jcol = g["valA"].copy(), g["jcol"].copy()
        p0 = g["irow"][r] - 1
        _swap(valA, p0, p0 + 1)
        _swap(jcol, p0, p0 + 1)
        assert E.probe_csr(valA, g["irow"], jcol).structured == 0


def test_an_entry_off_the_stencil_is_rejected(E):
    g = load_golden("g2_conducting_hole_16x15x14")
    sdz, sdy, sdx = g["geoPHYS"].shape
    nC = sdx * sdy * sdz
    irow, jcol = g["irow"], g["jcol"].copy()
    r = nC // 2                              # an interior Ax row: move its last column somewhere else
    p1 = irow[r + 1] - 2
    jcol[p1] = jcol[p1] + 3
    assert E.probe_csr(g["valA"], irow, jcol).structured == 0


def test_unrelated_matrices_are_rejected(E):
    from oracle import oracle as O
    valA, irow, jcol = O.poisson_csr(10, 10, 10)          # n = 1000: not three component blocks
    assert E.probe_csr(valA, irow, jcol).structured == 0
    n = 300
    irow = np.arange(1, n + 2, dtype=np.int32)            # diagonal matrix
    assert E.probe_csr(np.ones(n), irow, np.arange(1, n + 1, dtype=np.int32)).structured == 0


def test_single_component_box_in_three_blocks(E):
    """A one-component operator whose plane count is a multiple of 3 reads as three blocks of sdz/3 planes:
    the addresses are the same, so this is accepted (and gives such boxes the z-marching map)."""
    from oracle import oracle as O
    valA, irow, jcol = O.poisson_csr(8, 7, 15)
    p = E.probe_csr(valA, irow, jcol)
    assert p.structured == 1 and (p.sdx, p.sdy, p.sdz, p.n_cond) == (8, 7, 5, 0)
