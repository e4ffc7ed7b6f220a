"""VoxCad `.vxc` geometry ingest, writer and refiner.

What the reference does in /root/reference/src/vxc2data.f90:76-313 (scan the XML-ish text, decode one
`<Layer>` per z-plane, ZLIB+base64 or ASCII_READABLE) and :314-336, :604-636 (air split into domains of
500 000 cells, conductor cell numbering) — without the Python child process the reference spawns for
ZLIB inputs (src/uncompress_zlib.py, broken under numpy 2).  The palette `<Name>` strings carry the
reference's mini-language; this module reads what the operator needs from them (D, C, VEX/VEY/VEZ,
TRAN, SOLVER, BOUNDARY, which materials are sources) with an arithmetic-only evaluator.  Time functions
(FUNC ...), coil motion and the expression parser proper stay with the host program.
"""
from __future__ import annotations

import ast
import base64
import math
import re
import zlib
from dataclasses import dataclass, field

import numpy as np

# material id -> character of the ASCII_READABLE layer encoding (src/vxc2data.f90:71); '0' = air
LETTER = "123456789:;<=>?@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_`abcdefghijklmnopqrstuvwxyz"
MU0 = 0.12566370964050292e-05   # src/vxc2data.f90:402
E0 = 0.88541878176203908e-11    # :403


@dataclass
class VxcModel:
    vox: np.ndarray                      # uint8 [sdz, sdy, sdx], material id per voxel, 0 = air
    names: list                          # palette <Name> strings, material id = index + 1
    lattice_dim: float
    adj: tuple = (1.0, 1.0, 1.0)         # X/Y/Z_Dim_Adj
    compression: str = "ZLIB"
    extra: dict = field(default_factory=dict)

    @property
    def delta(self):
        return np.array([self.lattice_dim * a for a in self.adj])  # :103-124

    @property
    def shape_xyz(self):
        sdz, sdy, sdx = self.vox.shape
        return sdx, sdy, sdz


# --------------------------------------------------------------------------------------------- I/O
def read_vxc(path) -> VxcModel:
    txt = open(path, encoding="latin-1").read()

    def tag(t, default=None):
        m = re.search(rf"<{t}>(.*?)</{t}>", txt, flags=re.S)
        if m is None:
            if default is None:
                raise ValueError(f"{path}: <{t}> missing")
            return default
        return m.group(1).strip()

    sdx, sdy, sdz = int(tag("X_Voxels")), int(tag("Y_Voxels")), int(tag("Z_Voxels"))
    m = re.search(r'<Structure Compression="([A-Z_]+)"', txt)
    comp = m.group(1) if m else "ASCII_READABLE"
    layers = re.findall(r"<Layer><!\[CDATA\[(.*?)\]\]></Layer>", txt, flags=re.S)
    if len(layers) != sdz:
        raise ValueError(f"{path}: {len(layers)} layers for Z_Voxels = {sdz}")
    vox = np.zeros((sdz, sdy, sdx), np.uint8)
    for k, data in enumerate(layers):
        if comp == "ZLIB":
            raw = zlib.decompress(base64.b64decode(data))            # byte value = material id
            plane = np.frombuffer(raw, np.uint8)
        else:                                                        # index(letter, ch), :297-312
            plane = np.array([LETTER.find(c) + 1 for c in data.strip()], np.uint8)
        if plane.size != sdx * sdy:
            raise ValueError(f"{path}: layer {k} has {plane.size} voxels, expected {sdx * sdy}")
        vox[k] = plane.reshape(sdy, sdx)
    names = [s for s in re.findall(r"<Name>(.*?)</Name>", txt, flags=re.S)]
    num = lambda t: numeric(t[:10])   # CHARACTER(len=10) ch_e, src/vxc2data.f90:50: longer text is cut there
    return VxcModel(vox, names, num(tag("Lattice_Dim")),
                    (num(tag("X_Dim_Adj", "1")), num(tag("Y_Dim_Adj", "1")), num(tag("Z_Dim_Adj", "1"))),
                    comp)


def write_vxc(path, model: VxcModel, compression=None):
    comp = compression or model.compression
    sdz, sdy, sdx = model.vox.shape
    if max(sdx, sdy, sdz) > 999:
        raise ValueError("the reference reads the voxel counts with '(i3)': at most 999 per axis")
    for v in (model.lattice_dim, *model.adj):
        if len(repr(float(v))) > 10:
            raise ValueError(f"{v!r}: the reference reads only the first 10 characters of the lattice numbers "
                             "(src/vxc2data.f90:50); round it (vxc.resample does)")
    with open(path, "w", encoding="latin-1") as f:
        f.write('<?xml version="1.0" encoding="ISO-8859-1"?>\n<VXC Version="0.94">\n  <Lattice>\n')
        f.write(f"    <Lattice_Dim>{model.lattice_dim!r}</Lattice_Dim>\n")
        for ax, a in zip("XYZ", model.adj):
            f.write(f"    <{ax}_Dim_Adj>{a!r}</{ax}_Dim_Adj>\n")
        f.write("  </Lattice>\n  <Palette>\n")
        for i, nm in enumerate(model.names):
            f.write(f'    <Material ID="{i + 1}">\n      <Name>{nm}</Name>\n    </Material>\n')
        f.write(f'  </Palette>\n  <Structure Compression="{comp}">\n')
        f.write(f"    <X_Voxels>{sdx}</X_Voxels>\n    <Y_Voxels>{sdy}</Y_Voxels>\n    <Z_Voxels>{sdz}</Z_Voxels>\n")
        f.write("    <Data>\n")
        for k in range(sdz):
            plane = np.ascontiguousarray(model.vox[k]).reshape(-1)
            if comp == "ZLIB":
                data = base64.b64encode(zlib.compress(plane.tobytes())).decode()
            else:
                data = "".join("0" if v == 0 else LETTER[v - 1] for v in plane)
            f.write(f"      <Layer><![CDATA[{data}]]></Layer>\n")
        f.write("    </Data>\n  </Structure>\n</VXC>\n")


def refine(model: VxcModel, fx: int, fy: int, fz: int) -> VxcModel:
    """Every voxel becomes fx*fy*fz voxels; the physical size is kept (cell size / factor).  This is how
    the 256^3-class inputs of BASELINE configs 3 and 5 are made from the shipped 102x102x24 / 176x32x22."""
    vox = np.repeat(np.repeat(np.repeat(model.vox, fz, axis=0), fy, axis=1), fx, axis=2)
    # cut to the 10 characters the reference reads (see resample)
    adj = tuple(float(f"{a / f:.12g}"[:10]) for a, f in zip(model.adj, (fx, fy, fz)))
    return VxcModel(vox, list(model.names), model.lattice_dim, adj, model.compression)


def resample(model: VxcModel, sdx: int, sdy: int, sdz: int) -> VxcModel:
    """The same geometry on an sdx x sdy x sdz grid (any size, not only integer multiples): new voxel i takes
    the material of the old voxel that contains its centre, floor((i + 1/2) * old / new) per axis; the
    physical size is kept (cell size * old / new).  This is how the inputs of BASELINE configs 3 and 5
    (256x256x60 from the shipped 102x102x24, 384x192x128 from 176x32x22) are made.  For integer factors
    it is refine()."""
    oz, oy, ox = model.vox.shape
    if min(sdx, sdy, sdz) < 1:
        raise ValueError("resample: grid sizes must be positive")
    # integer arithmetic: floor((2 i + 1) * old / (2 new)), exact for every size
    ix = ((2 * np.arange(sdx, dtype=np.int64) + 1) * ox) // (2 * sdx)
    iy = ((2 * np.arange(sdy, dtype=np.int64) + 1) * oy) // (2 * sdy)
    iz = ((2 * np.arange(sdz, dtype=np.int64) + 1) * oz) // (2 * sdz)
    vox = np.ascontiguousarray(model.vox[np.ix_(iz, iy, ix)])
    # The reference keeps the text of Lattice_Dim and of the *_Dim_Adj values in a CHARACTER(len=10) variable
    # (src/vxc2data.f90:50, :97, :104): whatever a file says, it reads the first 10 characters.  So the factors
    # are cut to 10 characters here -- what write_vxc writes is then exactly what the reference reads.
    adj = tuple(float(f"{a * o / s:.12g}"[:10]) for a, o, s in zip(model.adj, (ox, oy, oz), (sdx, sdy, sdz)))
    return VxcModel(vox, list(model.names), model.lattice_dim, adj, model.compression)


# -------------------------------------------------------------------------- palette mini-language
_PREFIX = [("MEG", 1e6), ("PET", 1e15), ("M", 1e-3), ("K", 1e3), ("U", 1e-6), ("N", 1e-9), ("P", 1e-12),
           ("G", 1e9), ("T", 1e12), ("F", 1e-15), ("C", 1e-2), ("H", 1e2)]


def numeric(s: str) -> float:
    """SPICE-style number (src/utilites.f90:339-475): 5m, 0.4m, 1u, 10k, 1k3 (= 1.3k), 2meg, 1e-3."""
    t = s.strip().upper().replace(",", ".")
    try:
        return float(t)
    except ValueError:
        pass
    for p, mult in _PREFIX:
        i = t.find(p)
        if i > 0:
            head, tail = t[:i], t[i + len(p):]
            if tail and "." not in head:      # 1k3 -> 1.3k
                return float(head + "." + tail) * mult
            return float(head) * mult
    raise ValueError(f"not a number: {s!r}")


def evaluate(expr: str, consts: dict) -> float:
    """Quoted arithmetic of the palette ('mu0*35.26e6', '183/(6*dx*6*dz)'): + - * / ^ and the
    reference's constants (src/vxc2data.f90:398-412).  Function calls belong to the host's parser."""
    e = expr.strip()
    if e[:1] in "\"'`":
        e = e[1:-1]
    else:
        return numeric(e)
    tree = ast.parse(e.replace("^", "**").upper(), mode="eval")

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, (int, float)):
            return float(n.value)
        if isinstance(n, ast.Name):
            return float(consts[n.id])
        if isinstance(n, ast.UnaryOp) and isinstance(n.op, (ast.USub, ast.UAdd)):
            v = ev(n.operand)
            return -v if isinstance(n.op, ast.USub) else v
        if isinstance(n, ast.BinOp):
            a, b = ev(n.left), ev(n.right)
            if isinstance(n.op, ast.Add): return a + b
            if isinstance(n.op, ast.Sub): return a - b
            if isinstance(n.op, ast.Mult): return a * b
            if isinstance(n.op, ast.Div): return a / b
            if isinstance(n.op, ast.Pow): return a ** b
        raise NotImplementedError(f"expression {expr!r}: only arithmetic on constants is evaluated here")

    return ev(tree)


def _words(name: str):
    # '=' -> ' ', upper case, split on blanks (src/vxc2data.f90:131-142)
    return name.replace("=", " ").upper().split()


def domain_tables(model: VxcModel):
    """The tables gen_sparse_matrix and the time loop consume (src/vxc2data.f90:314-336, :442-636):
    geoPHYS, geoPHYS_C, valPHYS, BND, delta, dt, stop time, tol, itmax, and which materials are sources.
    Arrays are [sdz, sdy, sdx] (C order == the reference's (i,j,k) column-major)."""
    vox = model.vox
    sdz, sdy, sdx = vox.shape
    cells = vox.size
    nsub = int(vox.max())
    v = vox.reshape(-1).astype(np.int32).copy()
    # :316-333 -- the air cells in scan order, a new domain id every 500 000 cells: the p-th air cell (1-based) gets
    # nsub + 1 + p // 500000 (the counter is reset and the id advanced BEFORE the cell that completes a block is set)
    air = np.flatnonzero(v == 0)
    v[air] = nsub + 1 + np.arange(1, air.size + 1, dtype=np.int64) // 500000
    k = 1 + air.size // 500000
    if air.size % 500000 == 0:
        k -= 1
    nsub_glob = nsub + k
    if nsub_glob > 127:
        raise ValueError("more than 127 domains: geoPHYS is INTEGER(1) in the reference")
    delta = model.delta
    out = dict(tol=1e-3, itmax=10000, BND=np.full((3, 2), -0.95), dt=None, time=None, jump=None,
               solver="BCG", directory="out")     # :74 defaults
    for nm in model.names:                       # first pass: TRAN / SOLVER (:175-221)
        w = _words(nm)
        for i, word in enumerate(w):
            if word == "TRAN":
                for a, b in zip(w[i + 1::2], w[i + 2::2]):
                    if "STOP" in a: out["time"] = numeric(b)
                    elif "STEP" in a: out["dt"] = numeric(b)
                    elif "JUMP" in a: out["jump"] = numeric(b)
            elif word == "SOLVER":
                for a, b in zip(w[i + 1:], w[i + 2:]):
                    if "TOL" in a: out["tol"] = numeric(b)
                    elif "ITMAX" in a: out["itmax"] = int(round(numeric(b)))
                    elif "SOLV" in a: out["solver"] = b
                    elif "DIR" in a: out["directory"] = b
    consts = dict(PI=math.pi, E=0.27182818284590451e+001, MU0=MU0, E0=E0, DT=out["dt"] or 0.0, DX=delta[0],
                  DY=delta[1], DZ=delta[2], TIME=out["time"] or 0.0, NX=sdx, NY=sdy, NZ=sdz)
    valPHYS = np.zeros((nsub_glob, 5))
    valPHYS[nsub:, 0] = 1.0                      # air: D = 1 (:365-371)
    sources, conductors = {}, []
    for kp, nm in enumerate(model.names, start=1):
        w = nm.replace("=", " ").split()          # keep the case of quoted expressions' content irrelevant
        W = [x.upper() for x in w]
        for i in range(1, len(W)):
            if W[i][:1] == "D" and kp <= nsub and i + 1 < len(W) and not W[i].startswith("DIR"):
                valPHYS[kp - 1, 0] = evaluate(w[i + 1], consts)
                for jx in range(i + 2, len(W) - 1):
                    if W[jx][:1] == "C" and not W[jx].startswith("COS"):
                        valPHYS[kp - 1, 1] = evaluate(w[jx + 1], consts)
                        if valPHYS[kp - 1, 1] != 0.0:
                            conductors.append(kp)
                    elif "VEX" in W[jx]: valPHYS[kp - 1, 2] = evaluate(w[jx + 1], consts)
                    elif "VEY" in W[jx]: valPHYS[kp - 1, 3] = evaluate(w[jx + 1], consts)
                    elif "VEZ" in W[jx]: valPHYS[kp - 1, 4] = evaluate(w[jx + 1], consts)
                    elif W[jx] in ("SRCX", "SRCY", "SRCZ"):
                        sources[kp] = (W[jx][-1], W[jx + 1])
                break
            if "ENVIRON" in W[i]:                 # :571-592: properties of the LAST environment domain
                for jx in range(i + 1, len(W) - 1):
                    if W[jx][:1] == "D": valPHYS[nsub_glob - 1, 0] = evaluate(w[jx + 1], consts)
                    elif W[jx][:1] == "C":
                        valPHYS[nsub_glob - 1, 1] = evaluate(w[jx + 1], consts)
                        if valPHYS[nsub_glob - 1, 1] != 0.0:
                            conductors.append(nsub_glob)
                    elif "VEX" in W[jx]: valPHYS[nsub_glob - 1, 2] = evaluate(w[jx + 1], consts)
                    elif "VEY" in W[jx]: valPHYS[nsub_glob - 1, 3] = evaluate(w[jx + 1], consts)
                    elif "VEZ" in W[jx]: valPHYS[nsub_glob - 1, 4] = evaluate(w[jx + 1], consts)
            if "BOUNDARY" in W[i]:
                for a, b in zip(W[i + 1::2], w[i + 2::2]):
                    val = evaluate(b, consts)
                    key = a[:3]
                    if key == "ALL": out["BND"][:, :] = val
                    else:
                        d = "XYZ".index(key[1]); s_ = 0 if key[2] == "M" else 1
                        out["BND"][d, s_] = val
    geoPHYS = v.reshape(sdz, sdy, sdx).astype(np.int8)
    geoPHYS_C = np.zeros(cells, np.int32)         # :625-636: domain-major scan-order numbering
    m = 0
    for kp in conductors:
        idx = np.flatnonzero(v == kp)
        geoPHYS_C[idx] = 3 * cells + m + 1 + np.arange(idx.size)
        m += idx.size
    out.update(geoPHYS=geoPHYS, geoPHYS_C=geoPHYS_C.reshape(sdz, sdy, sdx), valPHYS=valPHYS, delta=delta,
               nsub=nsub, nsub_glob=nsub_glob, conductors=conductors, sources=sources, ncells0=m)
    return out
