"""ctypes binding of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; nothing under ``eddy_currents_3d_amd/`` does.  It is the checker, never the product.

* ``liboracle.so``            — our C restatement (oracle/ec3d_oracle.c) of
                                 /root/reference/src/solvers.f90:3-61 and src/EC3D.f90:465-1049.
* ``_ref/libref_solver.so``   — the unmodified reference solver compiled with amdflang
                                 (oracle/Makefile, target ``ref``); used to pin the restatement
                                 and as the ``"reference"`` CPU baseline.  Run out of process
                                 (``_ref/ref_solve``) because it keeps 48·n bytes on the stack.
"""
from __future__ import annotations

import ctypes as C
import os
import resource
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")


class GpuGeom(C.Structure):
    """Reduction geometry of the HIP kernels (ec3d_get_reduction_geometry)."""
    _fields_ = [("n_pad", C.c_int32), ("tile", C.c_int32), ("nblk", C.c_int32),
                ("threads", C.c_int32), ("xcd_group", C.c_int32), ("zm_tpp", C.c_int32),
                ("zm_pps", C.c_int32), ("ntiles_front", C.c_int32), ("ulist_n", C.c_int32),
                ("ulist", C.POINTER(C.c_int32)), ("visit_nwg", C.c_int32),
                ("visit_off", C.POINTER(C.c_int32)), ("visit", C.POINTER(C.c_int32)),
                ("patch_x", C.c_int32), ("patch_y", C.c_int32), ("patch_sdx", C.c_int32),
                ("patch_pitch", C.c_int32), ("patch_sdy", C.c_int32)]


def build(with_ref: bool = True) -> None:
    """Compile the checker (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", HERE, "all"], check=True)
    if with_ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-s", "-C", HERE, "ref"], check=True)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build(with_ref=False)
        L = C.CDLL(path)
        L.oracle_spmv_csr.argtypes = [_f64p, _i32p, _i32p, C.c_int32, _f64p, _f64p]
        L.oracle_spmv_csr.restype = None
        L.oracle_dot.argtypes = [_f64p, _f64p, C.c_int64]
        L.oracle_dot.restype = C.c_double
        L.oracle_norm2.argtypes = [_f64p, C.c_int64]
        L.oracle_norm2.restype = C.c_double
        L.oracle_bicgstab_wr.argtypes = [_f64p, _i32p, _i32p, C.c_int32, _f64p, _f64p, C.c_double,
                                         C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                         C.c_int32]
        L.oracle_bicgstab_wr.restype = C.c_int
        L.oracle_dot_gpuorder.argtypes = [C.POINTER(GpuGeom), _f64p, _f64p, C.c_int64]
        L.oracle_dot_gpuorder.restype = C.c_double
        L.oracle_dot_gpuorder_parts.argtypes = [C.POINTER(GpuGeom), _f64p, _f64p, C.c_int64, _f64p]
        L.oracle_dot_gpuorder_parts.restype = None
        L.oracle_tree_sum.argtypes = [_f64p, C.c_int32, C.c_int32]
        L.oracle_tree_sum.restype = C.c_double
        L.oracle_bicgstab_wr_gpuorder.argtypes = [C.POINTER(GpuGeom), C.POINTER(GpuGeom), _f64p, _i32p, _i32p, C.c_int32,
                                                  _f64p, _f64p, C.c_double, C.c_int32,
                                                  C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                                  C.c_int32]
        L.oracle_bicgstab_wr_gpuorder.restype = C.c_int
        L.oracle_bicgstab_wr_gpuorder3.argtypes = [C.POINTER(GpuGeom)] * 3 + [_f64p, _i32p, _i32p, C.c_int32,
                                                                                _f64p, _f64p, C.c_double, C.c_int32,
                                                                                C.POINTER(C.c_int32), C.c_void_p,
                                                                                C.c_void_p, C.c_int32]
        L.oracle_bicgstab_wr_gpuorder3.restype = C.c_int
        L.oracle_last_restart_count.argtypes = []
        L.oracle_last_restart_count.restype = C.c_int
        L.oracle_gen_sparse_matrix.argtypes = [C.c_int32, C.c_int32, C.c_int32, _i8p, _i32p, _f64p,
                                               C.c_int32, _f64p, _f64p, C.c_double, _i32p,
                                               C.c_void_p, C.c_void_p, C.POINTER(C.c_int64),
                                               C.POINTER(C.c_int32), C.c_void_p, _i32p]
        L.oracle_gen_sparse_matrix.restype = C.c_int
        L.oracle_poisson_csr.argtypes = [C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p, _i32p,
                                         C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.oracle_poisson_csr.restype = C.c_int
        L.oracle_poisson_csr_planes.argtypes = [C.c_int32, C.c_int32, C.c_int32, _f64p, _f64p, C.c_int32, C.c_int32,
                                                _i32p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.oracle_poisson_csr_planes.restype = C.c_int
        _lib = L
    return _lib


# --------------------------------------------------------------------------------------------
def spmv_csr(valA, irow, jcol, v):
    n = len(irow) - 1
    y = np.empty(n)
    lib().oracle_spmv_csr(valA, irow, jcol, n, np.ascontiguousarray(v, np.float64), y)
    return y


def dot(a, b):
    return lib().oracle_dot(np.ascontiguousarray(a), np.ascontiguousarray(b), len(a))


def norm2(a):
    return lib().oracle_norm2(np.ascontiguousarray(a), len(a))


def _hist(cap):
    hs = np.full(cap, np.nan)
    hr = np.full(cap, np.nan)
    return hs, hr


def bicgstab_wr(valA, irow, jcol, b, x0, tol, itmax, hist_cap=0):
    """Restatement of src/solvers.f90:3-50.  Returns (x, iter, hist_s, hist_r)."""
    n = len(irow) - 1
    x = np.array(x0, dtype=np.float64, copy=True)
    it = C.c_int32(0)
    hs, hr = _hist(hist_cap)
    lib().oracle_bicgstab_wr(valA, irow, jcol, n, np.ascontiguousarray(b, np.float64), x, tol, itmax,
                             C.byref(it), hs.ctypes.data, hr.ctypes.data, hist_cap)
    return x, it.value, hs, hr


def geoms_of(solver, whichs=(0, 1, 2)):
    """(vector-kernel geometry, SpMV-kernel geometry) of an eddy_currents_3d_amd.EC3DSolver."""
    out = []
    ul = np.ascontiguousarray(solver.ulist(), np.int32)
    for which in whichs:   # K4's grid, the SpMV kernels', K2's (3 .. 6: the split launches of a z-slab)
        g = solver.geometry(which)
        gg = GpuGeom(n_pad=g.n_pad, tile=g.tile, nblk=g.nblk, threads=g.threads, xcd_group=g.xcd_group,
                     zm_tpp=g.zm_tpp, zm_pps=g.zm_pps, ntiles_front=g.ntiles_front, ulist_n=g.ulist_n,
                     ulist=ul.ctypes.data_as(C.POINTER(C.c_int32)), patch_x=g.patch_x, patch_y=g.patch_y,
                     patch_sdx=g.patch_sdx, patch_pitch=getattr(g, "patch_pitch", 0),
                     patch_sdy=getattr(g, "patch_sdy", 0))
        gg._keep = ul  # the struct only holds a pointer
        if hasattr(solver, "visit_order"):   # the library's own account of its launches, tile by tile
            off, tiles = solver.visit_order(which)
            off = np.ascontiguousarray(off, np.int32)
            tiles = np.ascontiguousarray(tiles if len(tiles) else np.zeros(1, np.int32), np.int32)
            gg.visit_nwg = len(off) - 1
            gg.visit_off = off.ctypes.data_as(C.POINTER(C.c_int32))
            gg.visit = tiles.ctypes.data_as(C.POINTER(C.c_int32))
            gg._keep_visit = (off, tiles)
        out.append(gg)
    return tuple(out)


def bicgstab_wr_gpuorder(geom, valA, irow, jcol, b, x0, tol, itmax, hist_cap=0):
    """Same algorithm with the HIP kernels' summation order.  geom: (vector GpuGeom, SpMV GpuGeom)
    as returned by geoms_of(), or one GpuGeom used for both."""
    gv, gs = (geom[0], geom[1]) if isinstance(geom, tuple) else (geom, geom)
    gk2 = geom[2] if isinstance(geom, tuple) and len(geom) > 2 else gv
    n = len(irow) - 1
    x = np.array(x0, dtype=np.float64, copy=True)
    it = C.c_int32(0)
    hs, hr = _hist(hist_cap)
    lib().oracle_bicgstab_wr_gpuorder3(C.byref(gv), C.byref(gs), C.byref(gk2), valA, irow, jcol, n,
                                       np.ascontiguousarray(b, np.float64), x, tol, itmax,
                                       C.byref(it), hs.ctypes.data, hr.ctypes.data, hist_cap)
    return x, it.value, hs, hr


def device_system(solver, valA, irow, jcol, *vectors):
    """The reference's system in the DEVICE numbering of `solver` (identity unless it holds the structured
    A-V form, where U is embedded in the grid: rows without an unknown are empty).  Returns
    (valA, irow_dev, jcol_dev, row_map, [vectors in device numbering...]) for the GPU-order twin."""
    rm = solver.row_map().astype(np.int64)
    g = solver.geometry(0)
    n_dev = int(rm.max()) + 1 if len(rm) else 0
    n_dev = max(n_dev, min(int(g.n_pad), n_dev))
    lens = np.zeros(n_dev, np.int64)
    lens[rm] = np.diff(irow)
    irow_d = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    jcol_d = (rm[np.asarray(jcol, np.int64) - 1] + 1).astype(np.int32)   # rm is increasing: order kept
    out = []
    for v in vectors:
        d = np.zeros(n_dev)
        d[rm] = v
        out.append(d)
    return (valA, irow_d, jcol_d, rm, *out)


def twin_solve(solver, valA, irow, jcol, b, x0, tol, itmax, hist_cap=0):
    """The GPU-order twin on the system `solver` holds, whatever its device numbering; x comes back in
    the reference's numbering.  Rows the device adds (inactive U slots, plane padding) must stay zero."""
    vd, ird, jcd, rm, bd, xd = device_system(solver, valA, irow, jcol, b, x0)
    xo, it, hs, hr = bicgstab_wr_gpuorder(geoms_of(solver), vd, ird, jcd, bd, xd, tol, itmax, hist_cap=hist_cap)
    assert np.all(np.delete(xo, rm) == 0.0)
    return xo[rm], it, hs, hr


def tree_sum(vals, threads=256):
    """`vals` added the way the consumer kernels add a producer's partials or the ranks' sums (reduce_partials)."""
    v = np.ascontiguousarray(vals, np.float64)
    return lib().oracle_tree_sum(v, len(v), threads)


def dot_parts(geom, a, b):
    """Per-workgroup partial sums of one launch, workgroup order."""
    nwg = geom.visit_nwg if geom.visit_off else geom.nblk
    part = np.zeros(max(nwg, 1))
    lib().oracle_dot_gpuorder_parts(C.byref(geom), np.ascontiguousarray(a), np.ascontiguousarray(b), len(a), part)
    return part[:nwg]


def twin_solve_slabs(slabs, plan, valA, irow, jcol, b, x0, tol, itmax, hist_cap=0):
    """The GPU-order twin of a MULTI-RANK solve of the single-component operator: src/solvers.f90:3-50 on the whole
    system, every dot product summed as the z-slab drivers sum it -- per rank in that rank's kernels' order (the
    launches the rank's handle reports: ec3d_get_visit_order; a split kernel's partials are the first launch's
    followed by the second's), collapsed by the 256-thread tree (k_finalize), and the ranks' sums added in rank order
    by the same tree (reduce_partials over one value per rank).  slabs: [(EC3DSolver view of the slab, row0, row1)] in
    rank order, rows in the reference's numbering; plan: 0 plain, 1 K1 / K3 as interior + boundary launch, 2 K2 / K5 as
    boundary + interior launch, 3 three
    launches, 4 three launches with K4 and K5-in-K1 as boundary + interior launch, 5 = 1 and 2 together (what
    EC3DMulti.plan() reports).
    Returns (x, iter, hist_s, hist_r, restarts)."""
    spmv_w = (3, 4) if plan in (1, 5) else (1,)    # (plan 5: both splits at once)
    ss_w = (5, 6) if plan in (2, 5) else (2,)      # who sums S.S (plan 2: K2 as boundary + interior launch)
    k4_w = (7, 8) if plan == 4 else (0,)           # who sums R.R and R.R0
    k51_w = (7, 8) if plan == 4 else spmv_w        # who sums AP.R0 from the second iteration on (K5-in-K1 of the one before)
    geo = []
    for sv, lo, hi in slabs:
        g = {w: geoms_of(sv, (w,))[0] for w in set((0, 1) + ss_w + spmv_w + k4_w + k51_w)}
        geo.append((g, lo, hi))

    def dot(ws, u, v):
        vals = []
        for g, lo, hi in geo:
            parts = np.concatenate([dot_parts(g[w], u[lo:hi], v[lo:hi]) for w in ws])
            vals.append(tree_sum(parts))
        return tree_sum(vals)

    x = np.array(x0, dtype=np.float64, copy=True)
    hs, hr = _hist(hist_cap)
    R = b - spmv_csr(valA, irow, jcol, x)
    R0 = R.copy()
    P = R.copy()
    it = 0
    restarts = 0
    bnorm = np.sqrt(dot((1,), b, b))                         # k_residual
    if bnorm == 0.0:
        return x, 0, hs, hr, 0
    rr0 = dot((1,), R, R0)                                   # k_residual: R.R
    while True:
        if it > itmax:
            break
        it += 1
        AP = spmv_csr(valA, irow, jcol, P)
        alpha = rr0 / dot(spmv_w if it == 1 else k51_w, AP, R0)   # K1 (or K5-in-K1 of the previous iteration)
        S = R - alpha * AP
        nrm = np.sqrt(dot(ss_w, S, S))                       # K2 (or K2-in-K3)
        if it <= hist_cap:
            hs[it - 1] = nrm
        if nrm / bnorm < tol:
            x = x + alpha * P
            break
        AS = spmv_csr(valA, irow, jcol, S)
        omega = dot(spmv_w, AS, S) / dot(spmv_w, AS, AS)     # K3
        x = (x + alpha * P) + omega * S
        R = S - omega * AS
        rr = dot(k4_w, R, R)                                 # K4
        nrm = np.sqrt(rr)
        if it <= hist_cap:
            hr[it - 1] = nrm
        if nrm / bnorm < tol:
            break
        rr0_new = dot(k4_w, R, R0)                           # K4
        beta = (alpha / omega) * rr0_new / rr0
        P = R + beta * (P - omega * AP)
        if abs(rr0_new) / bnorm < tol:                       # src/solvers.f90:47-49
            R0 = R.copy()
            P = R.copy()
            restarts += 1
            rr0 = rr                                         # R0 == R: the next R.R0 is R.R in the same order
        else:
            rr0 = rr0_new
    return x, it, hs, hr, restarts


def last_restart_count():
    """Restarts (src/solvers.f90:47-49) the last twin solve took."""
    return int(lib().oracle_last_restart_count())


def dot_gpuorder(geom, a, b):
    return lib().oracle_dot_gpuorder(C.byref(geom), np.ascontiguousarray(a), np.ascontiguousarray(b),
                                     len(a))


def poisson_csr(sdx, sdy, sdz, delta=(0.00333, 0.00333, 0.00333), bnd=-0.95):
    """Config-2/4 operator: Ax block of a non-conducting box, 1-based CSR."""
    n = sdx * sdy * sdz
    BND = np.full(6, float(bnd)) if np.isscalar(bnd) else np.ascontiguousarray(bnd, np.float64)
    d = np.ascontiguousarray(delta, np.float64)
    irow = np.empty(n + 1, np.int32)
    nnz = C.c_int64(0)
    lib().oracle_poisson_csr(sdx, sdy, sdz, BND, d, irow, None, None, C.byref(nnz))
    jcol = np.empty(nnz.value, np.int32)
    valA = np.empty(nnz.value, np.float64)
    lib().oracle_poisson_csr(sdx, sdy, sdz, BND, d, irow, jcol.ctypes.data, valA.ctypes.data,
                             C.byref(nnz))
    return valA, irow, jcol


def poisson_rows_times(sdx, sdy, sdz, k0, k1, x, delta=(0.00333, 0.00333, 0.00333), bnd=-0.95):
    """Rows of the planes [k0, k1) (0-based) of poisson_csr(sdx, sdy, sdz) times the whole vector x: the CSR of
    those planes only (global columns), summed by the same oracle_spmv_csr (src/solvers.f90:54-61)."""
    BND = np.full(6, float(bnd)) if np.isscalar(bnd) else np.ascontiguousarray(bnd, np.float64)
    d = np.ascontiguousarray(delta, np.float64)
    rows = (k1 - k0) * sdx * sdy
    irow = np.empty(rows + 1, np.int32)
    nnz = C.c_int64(0)
    L = lib()
    L.oracle_poisson_csr_planes(sdx, sdy, sdz, BND, d, k0, k1, irow, None, None, C.byref(nnz))
    jcol = np.empty(nnz.value, np.int32)
    valA = np.empty(nnz.value, np.float64)
    L.oracle_poisson_csr_planes(sdx, sdy, sdz, BND, d, k0, k1, irow, jcol.ctypes.data, valA.ctypes.data, C.byref(nnz))
    x = np.ascontiguousarray(x, np.float64)
    assert x.size == sdx * sdy * sdz
    y = np.empty(rows)
    L.oracle_spmv_csr(valA, irow, jcol, rows, x, y)
    return y


def bar_rhs(N):
    """G5 / config-2 deterministic RHS: mu0*1e6 on the bar i,k in [N/2-2, N/2+3],
    j in [N/4, 3N/4] (1-based, inclusive), mu0 = the reference's constant
    (src/vxc2data.f90:402)."""
    mu0 = 0.12566370964050292e-05
    b = np.zeros((N, N, N))  # [k, j, i]
    lo, hi = N // 2 - 2, N // 2 + 3
    b[lo - 1:hi, N // 4 - 1:3 * N // 4, lo - 1:hi] = mu0 * 1e6
    return b.reshape(-1)


def gen_sparse_matrix(geoPHYS, geoPHYS_C, valPHYS, BND, delta, dt):
    """Restatement of src/EC3D.f90:465-1049.  Arrays are [k, j, i] C-order == Fortran (i,j,k).
    valPHYS: (nsub_glob, 5) array (row n-1 = domain n); BND: (3, 2) array BND[d, s].
    Returns dict(valA, irow, jcol, n, ncells0, cel_bnd=[X, Y, Z, Ux, Uy, Uz])."""
    sdz, sdy, sdx = geoPHYS.shape
    g = np.ascontiguousarray(geoPHYS, np.int8).reshape(-1)
    gc = np.ascontiguousarray(geoPHYS_C, np.int32).reshape(-1)
    vp = np.asarray(valPHYS, np.float64)
    nsub = vp.shape[0]
    vpf = np.ascontiguousarray(vp.T).reshape(-1)          # column-major valPHYS(n, c)
    bndf = np.ascontiguousarray(np.asarray(BND, np.float64).T).reshape(-1)  # BND(d, s)
    d = np.ascontiguousarray(delta, np.float64)
    ncell = sdx * sdy * sdz
    nc0 = int(np.count_nonzero(gc))
    n = 3 * ncell + nc0
    irow = np.zeros(n + 1, np.int32)
    nnz = C.c_int64(0)
    c0 = C.c_int32(0)
    nb = np.zeros(6, np.int32)
    rc = lib().oracle_gen_sparse_matrix(sdx, sdy, sdz, g, gc, vpf, nsub, bndf, d, dt, irow, None, None,
                                        C.byref(nnz), C.byref(c0), None, nb)
    if rc:
        raise RuntimeError(f"oracle_gen_sparse_matrix: reference would STOP (code {rc})")
    jcol = np.empty(nnz.value, np.int32)
    valA = np.empty(nnz.value, np.float64)
    lists = [np.empty(max(int(k), 1), np.int32) for k in nb]
    ptrs = (C.c_void_p * 6)(*[a.ctypes.data for a in lists])
    nb2 = np.zeros(6, np.int32)
    rc = lib().oracle_gen_sparse_matrix(sdx, sdy, sdz, g, gc, vpf, nsub, bndf, d, dt, irow,
                                        jcol.ctypes.data, valA.ctypes.data, C.byref(nnz),
                                        C.byref(c0), C.cast(ptrs, C.c_void_p), nb2)
    if rc:
        raise RuntimeError(f"oracle_gen_sparse_matrix: reference would STOP (code {rc})")
    return dict(valA=valA, irow=irow, jcol=jcol, n=n, ncells0=c0.value,
                cel_bnd=[a[:k] for a, k in zip(lists, nb2)])


# --------------------------------------------------------------------------------------------
# out-of-process solves (reference needs an unlimited stack: src/solvers.f90:11-12)
def have_ref() -> bool:
    return os.path.exists(os.path.join(REF_DIR, "ref_solve"))


def _unlimit_stack():
    try:
        resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
    except (ValueError, OSError):
        soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
        resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))


def count_sketch(x, m=4096, seed=0x9E3779B97F4A7C15):
    """Linear sketch of a long vector into m doubles: entry i goes to bucket h(i) with sign s(i) (splitmix64 of
    the index).  ||sketch(x) - sketch(y)||_2 estimates ||x - y||_2 without bias, relative standard deviation
    about sqrt(1/(2m)) (1.1 % at m = 4096): what lets a test state ||x_gpu - x_ref|| / ||x_ref|| for a vector far
    too large to commit.  Pure index arithmetic: the same buckets here and on the GPU box."""
    x = np.ascontiguousarray(x, np.float64).reshape(-1)
    out = np.zeros(m)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for lo in range(0, x.size, step):
            z = np.arange(lo, min(lo + step, x.size), dtype=np.uint64) + np.uint64(seed)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            bucket = ((z >> np.uint64(20)) % np.uint64(m)).astype(np.int64)
            sign = 1.0 - 2.0 * ((z >> np.uint64(7)) & np.uint64(1)).astype(np.float64)
            out += np.bincount(bucket, weights=sign * x[lo:lo + len(bucket)], minlength=m)
    return out


def solve_process(kind, valA, irow, jcol, b, x0, tol, itmax, nrep=1, capture_stdout=False):
    """Run ``kind`` in {"reference", "port"} as a child process on one core ("port_omp": the restatement under OpenMP on
    OMP_NUM_THREADS cores -- reporting only, see ec3d_oracle_omp.c).
    Returns (x, iter, seconds[, stdout])."""
    exe = (os.path.join(REF_DIR, "ref_solve") if kind == "reference" else
           os.path.join(HERE, "oracle_solve_omp") if kind == "port_omp" else os.path.join(HERE, "oracle_solve"))
    if not os.path.exists(exe):
        if kind in ("port", "port_omp"):
            build(with_ref=False)
        else:
            raise FileNotFoundError(exe)
    n = len(irow) - 1
    with tempfile.TemporaryDirectory(prefix="ec3d_oracle_") as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            np.array([n, len(jcol), itmax, nrep], np.int64).tofile(f)
            np.array([tol], np.float64).tofile(f)
            np.ascontiguousarray(irow, np.int32).tofile(f)
            np.ascontiguousarray(jcol, np.int32).tofile(f)
            np.ascontiguousarray(valA, np.float64).tofile(f)
            np.ascontiguousarray(b, np.float64).tofile(f)
            np.ascontiguousarray(x0, np.float64).tofile(f)
        p = subprocess.run([exe, fin, fout], preexec_fn=_unlimit_stack, check=True,
                           stdout=subprocess.PIPE if capture_stdout else None)
        with open(fout, "rb") as f:
            it = int(np.fromfile(f, np.int32, 2)[0])
            sec = float(np.fromfile(f, np.float64, 1)[0])
            x = np.fromfile(f, np.float64, n)
    if capture_stdout:
        return x, it, sec, p.stdout.decode()
    return x, it, sec
