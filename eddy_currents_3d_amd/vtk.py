"""Legacy-VTK field file exactly as the reference writes it (writeVtk_field,
/root/reference/src/utilites.f90:171-293): big-endian binary STRUCTURED_GRID with float32 points and
the point vectors Field_A, [Vector_field_eddy,] Vector_field_SOURCE, Vector_field_B.  The vectors come
from the device (EC3DSolver.vtk_fields); this module only formats the bytes.  Also the reference's second
file per output step, src_N.vtk (writeVtk_src, src/utilites.f90:3-168): the source cells as hexahedra."""
from __future__ import annotations

import os

import numpy as np


def _i8(*vals):
    # WRITE(buf,'(i8," ",i8," ",i8)') ... ; trim(adjustl(buf))   (utilites.f90:203-206)
    return " ".join("%8d" % v for v in vals).strip()


# One-entry caches, each a single (key, value) tuple that is replaced in ONE assignment: the writer and source threads of
# host._OutputPipeline call these functions side by side, and a dict that is cleared and refilled can lose its entry
# between one thread's store and its own lookup.  A thread that finds another key builds its value locally and returns
# that local; at worst two threads build the same block once.
_POINTS = (None, None)   # ((sdx, sdy, sdz, delta), the big-endian POINTS block): the same bytes in every file of a run


def _points_block(sdx, sdy, sdz, delta) -> bytes:
    global _POINTS
    key = (sdx, sdy, sdz, tuple(float(d) for d in delta))
    have_key, have = _POINTS
    if have_key != key:
        # REAL(k,8)*delta - delta, then REAL(.,4)   (utilites.f90:210-219)
        x = (np.arange(1, sdx + 1, dtype=np.float64) * delta[0] - delta[0]).astype(np.float32)
        y = (np.arange(1, sdy + 1, dtype=np.float64) * delta[1] - delta[1]).astype(np.float32)
        z = (np.arange(1, sdz + 1, dtype=np.float64) * delta[2] - delta[2]).astype(np.float32)
        pts = np.empty((sdz, sdy, sdx, 3), ">f4")
        pts[..., 0] = x[None, None, :]
        pts[..., 1] = y[None, :, None]
        pts[..., 2] = z[:, None, None]
        have = pts.tobytes()
        _POINTS = (key, have)                # one grid at a time: the block is 12 bytes per cell
    return have


def field_vtk_pieces(sdx, sdy, sdz, delta, fields):
    """The file as a list of bytes-like pieces, in order.  Vectors that already are big-endian float32 (dtype '>f4':
    what EC3DSolver.vtk_fields_wait hands out, swapped on the device) go out as views -- no copy, no conversion."""
    return _field_vtk_plan(sdx, sdy, sdz, delta, fields)


def _field_vtk_plan(sdx, sdy, sdz, delta, fields):
    n = sdx * sdy * sdz
    out = [b"# vtk DataFile Version 3.0\nout data result\nBINARY\n",
           ("DATASET STRUCTURED_GRID\nDIMENSIONS %s\n" % _i8(sdx, sdy, sdz)).encode(),
           ("POINTS %s float\n" % _i8(n)).encode(),
           _points_block(sdx, sdy, sdz, delta), b"\n", ("POINT_DATA %s\n" % _i8(n)).encode()]

    def vec(name, a):
        out = [("VECTORS %s float\n" % name).encode()]
        for part in (a if isinstance(a, (list, tuple)) else [a]):     # per-slab parts (EC3DMulti.vtk_fields_wait)
            if isinstance(part, _Hole):
                out.append(part)
                continue
            part = np.ascontiguousarray(part)
            if part.dtype != np.dtype(">f4"):
                part = part.astype(">f4")
            out.append(memoryview(part).cast("B"))
        return out + [b"\n"]

    out += vec("Field_A", fields["A"])
    if fields.get("eddy") is not None:
        out += vec("Vector_field_eddy", fields["eddy"])
    out += vec("Vector_field_SOURCE", fields["source"])
    out += vec("Vector_field_B", fields["B"])
    return out


class _Hole:
    """Stand-in for a vector of `nbytes` bytes when only the positions in the file are wanted."""

    def __init__(self, nbytes):
        self.nbytes = nbytes


def field_vtk_layout(sdx, sdy, sdz, delta, conducting):
    """Where everything of field_N.vtk lies: (text, data, size) -- text = [(offset, bytes)] for all that does not come
    from the device (header, POINTS block, the VECTORS lines, the newlines), data = {"A" | "eddy" | "source" | "B":
    offset of the vector's first byte}, size = the file's length.  Cell c of a vector is the 12 bytes at data[k] + 12 c:
    slabs of the grid are consecutive bytes, so every process of a multi-process run can put its own part where it
    belongs (host.run_slabs) and nothing has to be gathered."""
    n = sdx * sdy * sdz
    keys = ["A"] + (["eddy"] if conducting else []) + ["source", "B"]
    fields = {k: _Hole(12 * n) for k in keys}
    fields.setdefault("eddy", None)
    text, data, at = [], {}, 0
    holes = {id(v): k for k, v in fields.items() if v is not None}
    for p in _field_vtk_plan(sdx, sdy, sdz, delta, fields):
        if id(p) in holes:
            data[holes[id(p)]] = at
            at += p.nbytes
        else:
            text.append((at, p))
            at += len(p)
    return text, data, at


def join_parts(fields):
    """Per-slab lists (EC3DMulti.vtk_fields_wait) as one array per vector; arrays pass through."""
    return {k: (np.concatenate(v) if isinstance(v, (list, tuple)) else v) for k, v in fields.items()}


def field_vtk_bytes(sdx, sdy, sdz, delta, fields) -> bytes:
    return b"".join(bytes(p) for p in field_vtk_pieces(sdx, sdy, sdz, delta, fields))


def _write_pieces(path, pieces):
    """The pieces, in order, into `path`, unbuffered (the pieces are 100 MB views, not to be copied again).  One stream
    per file: buffered writes to one file serialise on its inode, so pwrite from several threads into the same file
    gained nothing (config 5: 6.8 GB/s with six threads per file); several FILES at a time do (host._OutputPipeline)."""
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        for p in pieces:
            mv = memoryview(p).cast("B")
            while len(mv):
                mv = mv[os.write(fd, mv):]
    finally:
        os.close(fd)


def write_field_vtk(path, sdx, sdy, sdz, delta, fields):
    _write_pieces(path, field_vtk_pieces(sdx, sdy, sdz, delta, fields))


_SRC_FIXED = (None, None)   # (ncell, (connectivity block, cell-type block)): they depend on the number of cells only


def src_vtk_pieces(sdx, sdy, sdz, delta, groups):
    """src_N.vtk (src/utilites.f90:3-168) as a list of bytes-like pieces: big-endian UNSTRUCTURED_GRID, one hexahedron
    (VTK type 11) per source cell with 8 double-precision corner points, and the cell vector Vector_field_SRC.
    groups: per source function, in the reference's order, (axis 0/1/2, cell ids (1-based, within one
    component), value) -- the cells where the function acts this step and its value (already times mu0).
    Written every output step for as many cells as the coils have (317 088 on config 5: 81 MB), so the corner
    coordinates go straight into one big-endian array and the blocks that depend on the cell count alone are kept."""
    ncell = sum(len(g[1]) for g in groups)
    d = np.asarray(delta, np.float64)
    corner = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]],
                      np.float64)
    pts = np.empty((ncell, 8, 3), ">f8")
    vec = np.zeros((ncell, 3), ">f8")
    at = 0
    for axis, cells, value in groups:                           # find_coord / write_coord, :107-166
        m = np.asarray(cells, np.int64) - 1
        k = len(m)
        ijk = np.empty((k, 1, 3), np.float64)
        ijk[:, 0, 0] = m % sdx + 1
        ijk[:, 0, 1] = (m // sdx) % sdy + 1
        ijk[:, 0, 2] = m // (sdx * sdy) + 1                     # 1-based i, j, k
        pts[at:at + k] = (ijk + corner[None, :, :]) * d - d     # REAL(i,8)*delta - delta
        vec[at:at + k, axis] = value
        at += k
    global _SRC_FIXED
    have_n, fixed = _SRC_FIXED
    if have_n != ncell:
        conn = np.empty((ncell, 9), ">i4")
        conn[:, 0] = 8
        conn[:, 1:] = 8 * np.arange(ncell)[:, None] + np.arange(8)[None, :]
        fixed = (conn.tobytes(), np.full(ncell, 11, ">i4").tobytes())
        _SRC_FIXED = (ncell, fixed)
    conn_b, types_b = fixed
    return [b"# vtk DataFile Version 3.0\nout data result\nBINARY\n", b"DATASET UNSTRUCTURED_GRID\n",
            ("POINTS %s double\n" % _i8(8 * ncell)).encode(), memoryview(pts).cast("B"), b"\n",
            ("CELLS %s %s\n" % (_i8(ncell), _i8(9 * ncell))).encode(), conn_b, b"\n",
            ("CELL_TYPES %s\n" % _i8(ncell)).encode(), types_b, b"\n", ("CELL_DATA %s\n" % _i8(ncell)).encode(),
            b"VECTORS Vector_field_SRC double\n", memoryview(vec).cast("B"), b"\n"]


def src_vtk_bytes(sdx, sdy, sdz, delta, groups) -> bytes:
    return b"".join(bytes(p) for p in src_vtk_pieces(sdx, sdy, sdz, delta, groups))


def write_src_vtk(path, sdx, sdy, sdz, delta, groups):
    _write_pieces(path, src_vtk_pieces(sdx, sdy, sdz, delta, groups))
