"""VTK field output against the files the unmodified reference wrote (tests/golden/*:
vtk_field_N = bytes of <DIR>/field_N.vtk, src/utilites.f90:171-293)."""
import numpy as np
import pytest

from conftest import load_golden


def parse_vectors(buf, n):
    """{name: float32 (n,3)} of every VECTORS block + the points, from a reference field_N.vtk."""
    out = {}
    key = b"POINTS "
    p = buf.index(key)
    p = buf.index(b"\n", p) + 1
    out["_points"] = np.frombuffer(buf[p:p + 12 * n], ">f4").reshape(n, 3)
    pos = p + 12 * n
    while True:
        q = buf.find(b"VECTORS ", pos)
        if q < 0:
            break
        e = buf.index(b"\n", q)
        name = buf[q + 8:e].split()[0].decode()
        out[name] = np.frombuffer(buf[e + 1:e + 1 + 12 * n], ">f4").reshape(n, 3)
        pos = e + 1 + 12 * n
    return out


@pytest.mark.parametrize("name", ["g1_nonconducting_8x7x6", "g2_conducting_hole_16x15x14"])
def test_vtk_formatter_reproduces_reference_bytes(name):
    """CPU: header, point coordinates and block layout -- re-emitting the reference's own vectors must
    give back its file byte for byte."""
    from eddy_currents_3d_amd.vtk import field_vtk_bytes
    g = load_golden(name)
    ref = g["vtk_field_1"].tobytes()
    sdz, sdy, sdx = g["geoPHYS"].shape
    v = parse_vectors(ref, sdx * sdy * sdz)
    fields = dict(A=v["Field_A"], eddy=v.get("Vector_field_eddy"), source=v["Vector_field_SOURCE"],
                  B=v["Vector_field_B"])
    assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], fields) == ref


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps", [("g1_nonconducting_8x7x6", [1]), ("g2_conducting_hole_16x15x14", [1]),
                                        ("g3_moving_coil_18x16x12", [1, 2])])
def test_device_fields_and_curl_reproduce_reference_file(name, steps, plane_pitch):
    """GPU: state right after the reference's solve k (x = its solution, b = its RHS) -> post_update on
    the device (src/EC3D.f90:412-433) -> ec3d_vtk_fields (curl A etc. on the device) -> file bytes
    identical to the reference's field_k.vtk."""
    import eddy_currents_3d_amd as E
    from eddy_currents_3d_amd.vtk import field_vtk_bytes
    g = load_golden(name)
    sdz, sdy, sdx = g["geoPHYS"].shape
    ncell = sdx * sdy * sdz
    conducting = bool(np.any(g["geoPHYS_C"] != 0))
    with E.EC3DSolver() as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        for k in steps:
            s.upload("X", g[f"xout{k}"])
            s.upload("B", g[f"b{k}"])
            s.post_update()
            f = s.vtk_fields(g["delta"], ncell, conducting)
            assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], f) == g[f"vtk_field_{k}"].tobytes()
            # the overlapped path (ec3d_vtk_fields_begin / _wait: side stream, pinned buffer, bytes swapped on the
            # device): the same file, from views that are written as they are
            slot = s.vtk_fields_begin(g["delta"], big_endian=True)
            fo = s.vtk_fields_wait(slot, big_endian=True)
            assert all(v is None or v.dtype == np.dtype(">f4") for v in fo.values())
            assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], fo) == g[f"vtk_field_{k}"].tobytes()
            slot = s.vtk_fields_begin(g["delta"], big_endian=False)           # the other buffer, host byte order
            fn = s.vtk_fields_wait(slot, big_endian=False)
            assert all(np.array_equal(fn[key], f[key]) for key in ("A", "source", "B"))
            assert (fn["eddy"] is None) == (f["eddy"] is None) and (f["eddy"] is None or np.array_equal(fn["eddy"], f["eddy"]))
        # a slot nothing was started in is refused, not handed out as whatever the buffer holds
        if len(steps) == 1:
            with pytest.raises(E.EC3DError, match="no such slot"):
                s.vtk_fields_wait(2)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_device_fields_on_slabs_reproduce_reference_file(world, plane_pitch):
    """The same on z-slabs (multi-GPU layout, slabs on this one GPU): each slab computes the fields of its
    owned planes, the curl reading the exchanged halo planes; concatenated -> the reference's file bytes."""
    import eddy_currents_3d_amd as E  # noqa: F401
    from eddy_currents_3d_amd.dist import HipAVSlabOps, InProcessSlabs, slab_bounds
    from eddy_currents_3d_amd.vtk import field_vtk_bytes
    g = load_golden("g3_moving_coil_18x16x12")
    sdz, sdy, sdx = g["geoPHYS"].shape
    ops = []
    for r in range(world):
        k0, k1 = slab_bounds(sdz, r, world)
        ops.append(HipAVSlabOps(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]),
                                k0, k1, world))
    drv = InProcessSlabs(ops)
    for k in (1, 2):
        for o in ops:
            o.set_vector_global("X", g[f"xout{k}"])
            o.set_vector_global("B", g[f"b{k}"])
        drv.post_update()
        f = drv.vtk_fields(g["delta"], True)
        assert field_vtk_bytes(sdx, sdy, sdz, g["delta"], f) == g[f"vtk_field_{k}"].tobytes()
    for o in ops:
        o.close()
