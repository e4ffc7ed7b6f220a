"""Legacy-VTK field file exactly as the reference writes it (writeVtk_field,
/root/reference/src/utilites.f90:171-293): big-endian binary STRUCTURED_GRID with float32 points and
the point vectors Field_A, [Vector_field_eddy,] Vector_field_SOURCE, Vector_field_B.  The vectors come
from the device (EC3DSolver.vtk_fields); this module only formats the bytes."""
from __future__ import annotations

import numpy as np


def _i8(*vals):
    # WRITE(buf,'(i8," ",i8," ",i8)') ... ; trim(adjustl(buf))   (utilites.f90:203-206)
    return " ".join("%8d" % v for v in vals).strip()


def field_vtk_bytes(sdx, sdy, sdz, delta, fields) -> bytes:
    n = sdx * sdy * sdz
    out = [b"# vtk DataFile Version 3.0\nout data result\nBINARY\n",
           ("DATASET STRUCTURED_GRID\nDIMENSIONS %s\n" % _i8(sdx, sdy, sdz)).encode(),
           ("POINTS %s float\n" % _i8(n)).encode()]
    # REAL(k,8)*delta - delta, then REAL(.,4)   (utilites.f90:210-219)
    x = (np.arange(1, sdx + 1, dtype=np.float64) * delta[0] - delta[0]).astype(np.float32)
    y = (np.arange(1, sdy + 1, dtype=np.float64) * delta[1] - delta[1]).astype(np.float32)
    z = (np.arange(1, sdz + 1, dtype=np.float64) * delta[2] - delta[2]).astype(np.float32)
    pts = np.empty((sdz, sdy, sdx, 3), np.float32)
    pts[..., 0] = x[None, None, :]
    pts[..., 1] = y[None, :, None]
    pts[..., 2] = z[:, None, None]
    out += [pts.astype(">f4").tobytes(), b"\n", ("POINT_DATA %s\n" % _i8(n)).encode()]

    def vec(name, a):
        return [("VECTORS %s float\n" % name).encode(), np.ascontiguousarray(a).astype(">f4").tobytes(), b"\n"]

    out += vec("Field_A", fields["A"])
    if fields.get("eddy") is not None:
        out += vec("Vector_field_eddy", fields["eddy"])
    out += vec("Vector_field_SOURCE", fields["source"])
    out += vec("Vector_field_B", fields["B"])
    return b"".join(out)


def write_field_vtk(path, sdx, sdy, sdz, delta, fields):
    with open(path, "wb") as f:
        f.write(field_vtk_bytes(sdx, sdy, sdz, delta, fields))
