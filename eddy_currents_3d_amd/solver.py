"""ctypes host over the C ABI of libec3d_hip.so (include/ec3d_hip.h).

Mirrors the reference's interface for the hot path:

* :func:`sprsBCGstabWR` — same name, argument order and meaning as
  ``SUBROUTINE sprsBCGstabWR (valA, irow, jcol, n, b, x, tolerance, itmax, iter)``
  (/root/reference/src/solvers.f90:3): 1-based CSR, ``x`` is updated in place (warm start),
  ``iter`` is returned.
* :class:`EC3DSolver` — the native handle API (assembly on the device, resident vectors,
  residual history, timing hooks).

No fallback: if the shared library or a HIP device is missing this raises :class:`EC3DError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("EC3D_LIB") or os.path.join(PKG, "libec3d_hip.so")   # EC3D_LIB: another build (A/B timing)


class EC3DError(RuntimeError):
    pass


class Geom(C.Structure):
    _fields_ = [("n_pad", C.c_int32), ("tile", C.c_int32), ("nblk", C.c_int32),
                ("threads", C.c_int32), ("xcd_group", C.c_int32), ("zm_tpp", C.c_int32),
                ("zm_pps", C.c_int32), ("ntiles_front", C.c_int32), ("ulist_n", C.c_int32),
                ("patch_x", C.c_int32), ("patch_y", C.c_int32), ("patch_sdx", C.c_int32),
                ("patch_pitch", C.c_int32), ("patch_sdy", C.c_int32)]


class MatrixInfo(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_pad", C.c_int64), ("nnz", C.c_int64), ("nbands", C.c_int32),
                ("band_offset", C.c_int32 * 16), ("tail_rows", C.c_int64),
                ("tail_entries_padded", C.c_int64), ("device_bytes", C.c_int64),
                ("dict_classes", C.c_int32)]


class CsrProbe(C.Structure):
    _fields_ = [("structured", C.c_int32), ("sdx", C.c_int32), ("sdy", C.c_int32), ("sdz", C.c_int32),
                ("n_cond", C.c_int32), ("classes", C.c_int32), ("plane_pitch", C.c_int32)]


VEC = dict(X=0, B=1, R=2, R0=3, P=4, AP=5, S=6, AS=7)
VTK_SLOTS = 3   # EC3D_VTK_SLOTS of include/ec3d_hip.h: pinned buffers of the overlapped field output
KERNEL = dict(spmv=0, k1=1, k2=2, k3=3, k4=4, k5=5)
# algorithmic bytes per row of each kernel with 7 bands
KERNEL_BYTES_PER_ROW = dict(spmv=72, k1=80, k2=24, k3=72, k4=56, k5=32)

EXPORTS = ["sprsbcgstabwr_", "ec3d_invalidate", "ec3d_create", "ec3d_destroy", "ec3d_last_error",
           "ec3d_set_matrix_csr", "ec3d_assemble", "ec3d_assemble_poisson", "ec3d_solve",
           "ec3d_upload", "ec3d_download", "ec3d_device_vector", "ec3d_solve_resident", "ec3d_spmv",
           "ec3d_export_csr", "ec3d_get_cel_bnd", "ec3d_get_reduction_geometry",
           "ec3d_set_workgroups", "ec3d_get_matrix_info", "ec3d_time_kernel", "ec3d_time_iterations",
           "ec3d_iterate_begin", "ec3d_iterate", "ec3d_get_fusion", "ec3d_get_x_interval", "ec3d_get_x_groups", "ec3d_get_k4_form", "ec3d_get_band_placement", "ec3d_get_vector_placement", "ec3d_place_vectors", "ec3d_set_format", "ec3d_set_stream",
           "ec3d_assemble_poisson_slab", "ec3d_vector_layout", "ec3d_adopt_vectors",
           "ec3d_dist_configure", "ec3d_dist_step", "ec3d_dist_set_boundary_rows", "ec3d_read_state_async", "ec3d_read_state", "ec3d_get_restart_count", "ec3d_set_zmarch", "ec3d_can_overlap",
           "ec3d_rhs_step", "ec3d_post_update", "ec3d_assemble_slab", "ec3d_vtk_fields", "ec3d_vtk_fields_begin", "ec3d_vtk_fields_wait",
           "ec3d_set_structured", "ec3d_get_row_map", "ec3d_get_ulist", "ec3d_probe_csr",
           "ec3d_device_synchronize", "ec3d_format_real8",
           "ec3d_multi_create", "ec3d_multi_destroy", "ec3d_multi_ranks", "ec3d_multi_slab", "ec3d_multi_set_format",
           "ec3d_multi_assemble_poisson", "ec3d_multi_assemble", "ec3d_multi_set_matrix_csr", "ec3d_multi_size",
           "ec3d_multi_upload", "ec3d_multi_download", "ec3d_multi_solve", "ec3d_multi_solve_resident",
           "ec3d_multi_rhs_step", "ec3d_multi_post_update", "ec3d_multi_vtk_fields", "ec3d_multi_vtk_fields_begin",
           "ec3d_multi_vtk_fields_wait", "ec3d_multi_iterate_begin",
           "ec3d_multi_iterate", "ec3d_multi_synchronize", "ec3d_true_residual", "ec3d_multi_true_residual", "ec3d_get_visit_order", "ec3d_probe_csr_multi", "ec3d_multi_spmv", "ec3d_multi_api_calls", "ec3d_multi_plan", "ec3d_multi_halo_rows", "ec3d_rccl_unique_id", "ec3d_multi_create_rank", "ec3d_format_real8_gfortran", "ec3d_multi_iterate_timed", "ec3d_multi_rccl_info"]

_f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32 = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i8 = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (soname
    libamdhip64.so.7, the same as /opt/rocm's).  If this library pulled in the system copy first and
    torch were imported later (dist.py, bench.py), the process would hold two runtimes and torch would
    see no GPU.  So when torch is installed but not imported yet, load ITS runtime first; our DT_NEEDED
    then resolves to it by soname.  EC3D_HIP_RUNTIME=system keeps /opt/rocm's (torch-free processes)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("EC3D_HIP_RUNTIME") == "system":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libec3d_hip.so and declare every prototype of include/ec3d_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIBPATH
    if not os.path.exists(p):
        raise EC3DError(f"{p} not built: run `python -m eddy_currents_3d_amd.build` "
                        "(there is no CPU fallback)")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    if os.environ.get("EC3D_LIB"):
        # another (older) build for a same-box A/B timing (tools/ab_perf.py): entry points added since are absent
        # there; calling one raises AttributeError, declaring it must not
        class _Tolerant:
            def __init__(self, lib):
                object.__setattr__(self, "_lib", lib)

            def __getattr__(self, name):
                try:
                    return getattr(self._lib, name)
                except AttributeError:
                    class _Missing:
                        argtypes = restype = None

                        def __call__(self, *a):
                            raise EC3DError(f"{p} does not export {name}")
                    return _Missing()
        L = _Tolerant(L)
    hp = C.c_void_p
    L.ec3d_last_error.restype = C.c_char_p
    L.ec3d_create.argtypes = [C.POINTER(hp), C.c_int]
    L.ec3d_destroy.argtypes = [hp]
    L.ec3d_set_matrix_csr.argtypes = [hp, C.c_int32, _f64, _i32, _i32]
    L.ec3d_assemble.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, _i8, _i32, _f64, C.c_int32, _f64,
                                _f64, C.c_double]
    L.ec3d_assemble_poisson.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, _f64, _f64]
    L.ec3d_solve.argtypes = [hp, _f64, _f64, C.c_double, C.c_int32, C.POINTER(C.c_int32), hp, C.c_int32]
    L.ec3d_upload.argtypes = [hp, C.c_int, _f64]
    L.ec3d_download.argtypes = [hp, C.c_int, _f64]
    L.ec3d_device_vector.argtypes = [hp, C.c_int, C.POINTER(hp), C.POINTER(C.c_int64)]
    L.ec3d_solve_resident.argtypes = [hp, C.c_double, C.c_int32, C.POINTER(C.c_int32), hp, C.c_int32]
    L.ec3d_spmv.argtypes = [hp, _f64, _f64]
    L.ec3d_export_csr.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int64), hp, hp, hp]
    L.ec3d_get_cel_bnd.argtypes = [hp, C.c_int, C.POINTER(C.c_int32), hp]
    L.ec3d_get_reduction_geometry.argtypes = [hp, C.c_int, C.POINTER(Geom)]
    L.ec3d_set_zmarch.argtypes = [hp, C.c_int]
    L.ec3d_can_overlap.argtypes = [hp]
    L.ec3d_assemble_slab.argtypes = [hp] + [C.c_int32] * 7 + [_i8, _i32, _f64, C.c_int32, _f64, _f64, C.c_double]
    L.ec3d_rhs_step.argtypes = [hp, C.c_int32, C.c_int32, _i32, _f64]
    L.ec3d_post_update.argtypes = [hp]
    L.ec3d_vtk_fields.argtypes = [hp, _f64, hp, hp, hp, hp]
    L.ec3d_vtk_fields_begin.argtypes = [hp, _f64, C.c_int32, C.POINTER(C.c_int32)]
    L.ec3d_vtk_fields_wait.argtypes = [hp, C.c_int32] + [C.POINTER(C.POINTER(C.c_float))] * 4 + [C.POINTER(C.c_int64)]
    L.ec3d_set_workgroups.argtypes = [hp, C.c_int32]
    L.ec3d_get_matrix_info.argtypes = [hp, C.POINTER(MatrixInfo)]
    L.ec3d_time_kernel.argtypes = [hp, C.c_int, C.c_int32, C.POINTER(C.c_double)]
    L.ec3d_time_iterations.argtypes = [hp, C.c_int32, C.POINTER(C.c_double)]
    L.ec3d_device_synchronize.argtypes = [hp]
    L.ec3d_format_real8.argtypes = [C.c_double, C.c_char_p]
    L.ec3d_format_real8.restype = None
    L.ec3d_iterate_begin.argtypes = [hp]
    L.ec3d_iterate.argtypes = [hp, C.c_int32, C.c_int32, hp]
    L.ec3d_get_fusion.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ec3d_get_x_interval.argtypes = [hp, C.POINTER(C.c_int32)]
    L.ec3d_get_x_groups.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ec3d_get_k4_form.argtypes = [hp, C.POINTER(C.c_int32)]
    L.ec3d_get_band_placement.argtypes = [hp, C.c_int32, hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ec3d_get_vector_placement.argtypes = [hp, C.c_int32, hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.ec3d_place_vectors.argtypes = [hp, C.c_int32]
    L.ec3d_set_format.argtypes = [hp, C.c_int]
    L.ec3d_set_structured.argtypes = [hp, C.c_int]
    L.ec3d_get_row_map.argtypes = [hp, _i32]
    L.ec3d_dist_set_boundary_rows.argtypes = [hp, C.c_int32, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"),
                                              np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS"),
                                              C.POINTER(C.c_int32)]
    L.ec3d_probe_csr.argtypes = [C.c_int32, _f64, _i32, _i32, C.POINTER(CsrProbe)]
    L.ec3d_probe_csr_multi.argtypes = [C.c_int32, _f64, _i32, _i32, C.c_int32, C.POINTER(C.c_int32)]
    L.ec3d_get_ulist.argtypes = [hp, _i32]
    L.ec3d_set_stream.argtypes = [hp, hp]
    L.ec3d_assemble_poisson_slab.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _f64, _f64]
    L.ec3d_vector_layout.argtypes = [hp] + [C.POINTER(C.c_int64)] * 4
    L.ec3d_adopt_vectors.argtypes = [hp, hp]
    L.ec3d_dist_configure.argtypes = [hp, C.c_int32, hp, hp]
    L.ec3d_dist_step.argtypes = [hp, C.c_int32, C.c_int32, C.c_double]
    L.ec3d_read_state_async.argtypes = [hp, C.c_void_p]
    L.ec3d_read_state.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.ec3d_get_restart_count.argtypes = [hp, C.POINTER(C.c_int32)]
    L.ec3d_get_visit_order.argtypes = [hp, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64), hp, hp]
    L.ec3d_true_residual.argtypes = [hp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ec3d_multi_true_residual.argtypes = [hp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.ec3d_multi_create.argtypes = [C.POINTER(hp), C.c_int32, hp]
    L.ec3d_multi_destroy.argtypes = [hp]
    L.ec3d_multi_ranks.argtypes = [hp]
    L.ec3d_multi_slab.argtypes = [hp, C.c_int32, C.POINTER(hp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ec3d_multi_set_format.argtypes = [hp, C.c_int, C.c_int]
    L.ec3d_multi_assemble_poisson.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, _f64, _f64]
    L.ec3d_multi_assemble.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, _i8, _i32, _f64, C.c_int32, _f64,
                                      _f64, C.c_double]
    L.ec3d_multi_set_matrix_csr.argtypes = [hp, C.c_int32, _f64, _i32, _i32]
    L.ec3d_multi_size.argtypes = [hp, C.POINTER(C.c_int64)]
    L.ec3d_multi_upload.argtypes = [hp, C.c_int, _f64]
    L.ec3d_multi_download.argtypes = [hp, C.c_int, _f64]
    L.ec3d_multi_solve.argtypes = [hp, _f64, _f64, C.c_double, C.c_int32, C.POINTER(C.c_int32)]
    L.ec3d_multi_solve_resident.argtypes = [hp, C.c_double, C.c_int32, C.POINTER(C.c_int32)]
    L.ec3d_multi_rhs_step.argtypes = [hp, C.c_int32, C.c_int32, _i32, _f64]
    L.ec3d_multi_post_update.argtypes = [hp]
    L.ec3d_multi_spmv.argtypes = [hp, _f64, _f64]
    L.ec3d_multi_vtk_fields.argtypes = [hp, _f64, hp, hp, hp, hp]
    L.ec3d_multi_vtk_fields_begin.argtypes = [hp, _f64, C.c_int32, C.POINTER(C.c_int32)]
    L.ec3d_multi_vtk_fields_wait.argtypes = ([hp, C.c_int32, C.c_int32] + [C.POINTER(C.POINTER(C.c_float))] * 4 +
                                             [C.POINTER(C.c_int64)] * 2)
    L.ec3d_multi_iterate_begin.argtypes = [hp]
    L.ec3d_multi_iterate.argtypes = [hp, C.c_int32, C.c_int32, hp]
    L.ec3d_multi_synchronize.argtypes = [hp]
    L.ec3d_multi_api_calls.argtypes = [hp, C.c_int32, C.POINTER(C.c_double)]
    L.ec3d_multi_iterate_timed.argtypes = [hp, C.c_int32, C.c_int32, C.c_int32, hp, hp, hp]
    L.ec3d_multi_rccl_info.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    L.ec3d_multi_plan.argtypes = [hp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ec3d_multi_halo_rows.argtypes = [hp, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.ec3d_rccl_unique_id.argtypes = [C.c_char_p]
    L.ec3d_multi_create_rank.argtypes = [C.POINTER(hp), C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_char_p, C.c_int32,
                                         C.c_int32]
    L.sprsbcgstabwr_.argtypes = [_f64, _i32, _i32, C.POINTER(C.c_int32), _f64, _f64,
                                 C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.sprsbcgstabwr_.restype = None
    L.ec3d_invalidate.restype = None
    if path is None:
        _lib = L
    return L


def _chk(L, rc, what):
    if rc != 0:
        raise EC3DError(f"{what} failed ({rc}): {L.ec3d_last_error().decode()}")


def probe_csr(valA, irow, jcol):
    """Host-only (no GPU): how would the library store this CSR matrix?  Returns a CsrProbe; `.structured`
    says whether the class-coded A-V form applies (include/ec3d_hip.h, ec3d_probe_csr)."""
    L = load_library()
    out = CsrProbe()
    irow = np.ascontiguousarray(irow, np.int32)
    rc = L.ec3d_probe_csr(len(irow) - 1, np.ascontiguousarray(valA, np.float64), irow,
                          np.ascontiguousarray(jcol, np.int32), C.byref(out))
    if rc:
        raise EC3DError(f"ec3d_probe_csr failed ({rc})")
    return out


def probe_csr_multi(valA, irow, jcol, nranks: int):
    """Host-only: (cuttable, reason) -- would the library cut this CSR matrix into `nranks` z-slabs
    (EC3DMulti.set_matrix_csr, the drop-in symbol under EC3D_NGPU)?"""
    L = load_library()
    ok = C.c_int32(0)
    irow = np.ascontiguousarray(irow, np.int32)
    rc = L.ec3d_probe_csr_multi(len(irow) - 1, np.ascontiguousarray(valA, np.float64), irow,
                                np.ascontiguousarray(jcol, np.int32), int(nranks), C.byref(ok))
    if rc:
        raise EC3DError(f"ec3d_probe_csr_multi failed ({rc})")
    return bool(ok.value), ("" if ok.value else L.ec3d_last_error().decode())


def sprsBCGstabWR(valA, irow, jcol, n, b, x, tolerance, itmax):
    """Drop-in for the reference solver (src/solvers.f90:3, called at src/EC3D.f90:408).

    ``valA`` f64[nnz], ``irow`` i32[n+1] and ``jcol`` i32[nnz] are the reference's 1-based CSR;
    ``x`` (f64[n], C-contiguous) is the warm start and receives the solution in place.
    Returns ``iter``.  Goes through the exported F77 symbol ``sprsbcgstabwr_`` itself."""
    L = load_library()
    if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous):
        raise TypeError("x must be a C-contiguous float64 ndarray (updated in place)")
    it = C.c_int32(0)
    L.sprsbcgstabwr_(np.ascontiguousarray(valA, np.float64), np.ascontiguousarray(irow, np.int32),
                     np.ascontiguousarray(jcol, np.int32), C.byref(C.c_int32(int(n))),
                     np.ascontiguousarray(b, np.float64), x, C.byref(C.c_double(float(tolerance))),
                     C.byref(C.c_int32(int(itmax))), C.byref(it))
    return it.value


class EC3DSolver:
    """Handle API of include/ec3d_hip.h (one HIP device, one stream)."""

    def __init__(self, device: int = 0, nblk: int | None = None, dictionary: bool | None = None,
                 structured: bool | None = None):
        self.L = load_library()
        self.h = C.c_void_p()
        _chk(self.L, self.L.ec3d_create(C.byref(self.h), device), "ec3d_create")
        if nblk:
            self.set_workgroups(nblk)
        if dictionary is not None:
            self.set_format(dictionary)
        if structured is not None:
            _chk(self.L, self.L.ec3d_set_structured(self.h, int(bool(structured))), "ec3d_set_structured")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.ec3d_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- matrix ---------------------------------------------------------------------------
    def set_matrix_csr(self, valA, irow, jcol):
        irow = np.ascontiguousarray(irow, np.int32)
        _chk(self.L, self.L.ec3d_set_matrix_csr(self.h, len(irow) - 1, np.ascontiguousarray(valA, np.float64),
                                                irow, np.ascontiguousarray(jcol, np.int32)),
             "ec3d_set_matrix_csr")

    def assemble(self, geoPHYS, geoPHYS_C, valPHYS, BND, delta, dt):
        """geoPHYS/geoPHYS_C: [sdz, sdy, sdx] (C order == Fortran (i,j,k)); valPHYS (nsub_glob, 5);
        BND (3, 2).  Replaces gen_sparse_matrix (src/EC3D.f90:465-1049)."""
        sdz, sdy, sdx = geoPHYS.shape
        vp = np.asarray(valPHYS, np.float64)
        _chk(self.L, self.L.ec3d_assemble(
            self.h, sdx, sdy, sdz, np.ascontiguousarray(geoPHYS, np.int8).reshape(-1),
            np.ascontiguousarray(geoPHYS_C, np.int32).reshape(-1),
            np.ascontiguousarray(vp.T).reshape(-1), vp.shape[0],
            np.ascontiguousarray(np.asarray(BND, np.float64).T).reshape(-1),
            np.ascontiguousarray(delta, np.float64), float(dt)), "ec3d_assemble")

    def assemble_slab(self, sdz_global, e0, e1, k0, k1, geoPHYS_ext, geoPHYS_C_ext, valPHYS, BND, delta, dt):
        """One z-slab of the A-V system on the extended grid [e0, e1) (owned planes [k0, k1) + 2 halo
        planes per interior side); geoPHYS_C_ext numbers the extended slab's conducting cells locally."""
        nz, sdy, sdx = geoPHYS_ext.shape
        assert nz == e1 - e0
        vp = np.asarray(valPHYS, np.float64)
        _chk(self.L, self.L.ec3d_assemble_slab(
            self.h, sdx, sdy, sdz_global, e0, e1, k0, k1,
            np.ascontiguousarray(geoPHYS_ext, np.int8).reshape(-1),
            np.ascontiguousarray(geoPHYS_C_ext, np.int32).reshape(-1),
            np.ascontiguousarray(vp.T).reshape(-1), vp.shape[0],
            np.ascontiguousarray(np.asarray(BND, np.float64).T).reshape(-1),
            np.ascontiguousarray(delta, np.float64), float(dt)), "ec3d_assemble_slab")

    def ulist(self):
        """Occupied tiles of the U block (structured A-V form), in the order the kernels visit them."""
        k = self.geometry(0).ulist_n
        t = np.zeros(max(k, 1), np.int32)
        _chk(self.L, self.L.ec3d_get_ulist(self.h, t), "ec3d_get_ulist")
        return t[:k]

    def row_map(self):
        """Device row of every unknown of the reference's numbering (identity unless structured)."""
        m = np.empty(self.n, np.int32)
        _chk(self.L, self.L.ec3d_get_row_map(self.h, m), "ec3d_get_row_map")
        return m

    def set_format(self, dictionary: bool):
        """True (default): 1 class byte per row + coefficient table when the operator allows it;
        False: plain DIA coefficient streams.  Call before assembling / setting the matrix."""
        _chk(self.L, self.L.ec3d_set_format(self.h, int(bool(dictionary))), "ec3d_set_format")

    def assemble_poisson(self, sdx, sdy, sdz, delta=(0.00333, 0.00333, 0.00333), bnd=-0.95, slab=None):
        """Non-conducting Ax block (src/EC3D.f90:528-654).  slab=(k0, k1): only planes [k0, k1)."""
        BND = np.full(6, float(bnd)) if np.isscalar(bnd) else np.ascontiguousarray(
            np.asarray(bnd, np.float64).T).reshape(-1)
        d = np.ascontiguousarray(delta, np.float64)
        if slab is None:
            _chk(self.L, self.L.ec3d_assemble_poisson(self.h, sdx, sdy, sdz, BND, d), "ec3d_assemble_poisson")
        else:
            _chk(self.L, self.L.ec3d_assemble_poisson_slab(self.h, sdx, sdy, sdz, int(slab[0]), int(slab[1]),
                                                           BND, d), "ec3d_assemble_poisson_slab")

    def export_csr(self):
        n, nnz = C.c_int32(0), C.c_int64(0)
        _chk(self.L, self.L.ec3d_export_csr(self.h, C.byref(n), C.byref(nnz), None, None, None), "ec3d_export_csr")
        irow = np.empty(n.value + 1, np.int32)
        jcol = np.empty(nnz.value, np.int32)
        valA = np.empty(nnz.value, np.float64)
        _chk(self.L, self.L.ec3d_export_csr(self.h, C.byref(n), C.byref(nnz), irow.ctypes.data,
                                            jcol.ctypes.data, valA.ctypes.data), "ec3d_export_csr")
        return valA, irow, jcol

    def cel_bnd(self):
        out = []
        for w in range(6):
            k = C.c_int32(0)
            _chk(self.L, self.L.ec3d_get_cel_bnd(self.h, w, C.byref(k), None), "ec3d_get_cel_bnd")
            a = np.empty(max(k.value, 1), np.int32)
            _chk(self.L, self.L.ec3d_get_cel_bnd(self.h, w, C.byref(k), a.ctypes.data), "ec3d_get_cel_bnd")
            out.append(a[:k.value])
        return out

    @property
    def info(self) -> MatrixInfo:
        mi = MatrixInfo()
        _chk(self.L, self.L.ec3d_get_matrix_info(self.h, C.byref(mi)), "ec3d_get_matrix_info")
        return mi

    @property
    def n(self) -> int:
        return int(self.info.n)

    def geometry(self, which: int = 0) -> Geom:
        """Reduction geometry of the vector kernels (which=0) or of the SpMV kernels (which=1)."""
        g = Geom()
        _chk(self.L, self.L.ec3d_get_reduction_geometry(self.h, which, C.byref(g)), "ec3d_get_reduction_geometry")
        return g

    def visit_order(self, which: int = 0):
        """(offsets[nwg + 1], tiles): the 512-row tiles every workgroup of the vector kernels (0) / SpMV kernels
        (1) visits, in order -- the summation order of the dot products."""
        nwg, tot = C.c_int32(0), C.c_int64(0)
        _chk(self.L, self.L.ec3d_get_visit_order(self.h, which, C.byref(nwg), C.byref(tot), None, None),
             "ec3d_get_visit_order")
        off = np.zeros(nwg.value + 1, np.int32)
        tiles = np.zeros(max(tot.value, 1), np.int32)
        _chk(self.L, self.L.ec3d_get_visit_order(self.h, which, C.byref(nwg), C.byref(tot), off.ctypes.data,
                                                 tiles.ctypes.data), "ec3d_get_visit_order")
        return off, tiles[:tot.value]

    def set_zmarch(self, on: bool):
        _chk(self.L, self.L.ec3d_set_zmarch(self.h, int(bool(on))), "ec3d_set_zmarch")

    def set_workgroups(self, nblk: int):
        _chk(self.L, self.L.ec3d_set_workgroups(self.h, int(nblk)), "ec3d_set_workgroups")

    # ---- solve ----------------------------------------------------------------------------
    def solve(self, b, x0, tolerance, itmax, hist_cap: int = 0):
        """One reference solve (src/solvers.f90:3-50).  Returns (x, iter, hist[hist_cap, 2])."""
        x = np.array(x0, dtype=np.float64, copy=True)
        it = C.c_int32(0)
        hist = np.full((max(hist_cap, 1), 2), np.nan)
        _chk(self.L, self.L.ec3d_solve(self.h, np.ascontiguousarray(b, np.float64), x, float(tolerance),
                                       int(itmax), C.byref(it), hist.ctypes.data if hist_cap else None,
                                       hist_cap), "ec3d_solve")
        return x, it.value, hist[:hist_cap]

    def upload(self, which: str, a):
        _chk(self.L, self.L.ec3d_upload(self.h, VEC[which], np.ascontiguousarray(a, np.float64)), "ec3d_upload")

    def download(self, which: str):
        a = np.empty(self.n)
        _chk(self.L, self.L.ec3d_download(self.h, VEC[which], a), "ec3d_download")
        return a

    def device_vector(self, which: str):
        p, n = C.c_void_p(), C.c_int64(0)
        _chk(self.L, self.L.ec3d_device_vector(self.h, VEC[which], C.byref(p), C.byref(n)), "ec3d_device_vector")
        return p.value, n.value

    def solve_resident(self, tolerance, itmax, hist_cap: int = 0):
        it = C.c_int32(0)
        hist = np.full((max(hist_cap, 1), 2), np.nan)
        _chk(self.L, self.L.ec3d_solve_resident(self.h, float(tolerance), int(itmax), C.byref(it),
                                                hist.ctypes.data if hist_cap else None, hist_cap),
             "ec3d_solve_resident")
        return it.value, hist[:hist_cap]

    # ---- time-loop field work on the resident vectors (src/EC3D.f90:275-404, :412-433) --------
    def rhs_step(self, src_index, src_value, moving: bool = False):
        """Jaf for this step: source scatter (1-based unknown ids, values from the host's source
        functions), inertial terms, U-row right-hand sides, cel_bnd* zero-fills."""
        idx = np.ascontiguousarray(src_index, np.int32)
        val = np.ascontiguousarray(src_value, np.float64)
        _chk(self.L, self.L.ec3d_rhs_step(self.h, int(bool(moving)), len(idx), idx if len(idx) else np.zeros(1, np.int32),
                                          val if len(val) else np.zeros(1)), "ec3d_rhs_step")

    def post_update(self):
        _chk(self.L, self.L.ec3d_post_update(self.h), "ec3d_post_update")

    def vtk_fields(self, delta, ncells: int, conducting: bool, zero_eddy: bool = False):
        """float32 point vectors of field_N.vtk (src/utilites.f90:222-289) from the resident X, B.
        Returns dict(A, eddy (None without conductors), source, B), each (ncells, 3); ncells = the cells the
        handle owns.  zero_eddy: start the eddy field from zeros (a z-slab that holds no conductor)."""
        mk = lambda: np.empty((ncells, 3), np.float32)
        fa, fs, fb = mk(), mk(), mk()
        fe = (np.zeros((ncells, 3), np.float32) if zero_eddy else mk()) if conducting else None
        _chk(self.L, self.L.ec3d_vtk_fields(self.h, np.ascontiguousarray(delta, np.float64), fa.ctypes.data,
                                            fe.ctypes.data if conducting else None, fs.ctypes.data,
                                            fb.ctypes.data), "ec3d_vtk_fields")
        return dict(A=fa, eddy=fe, source=fs, B=fb)

    def vtk_fields_begin(self, delta, big_endian: bool = True) -> int:
        """Start the field output of this step WITHOUT waiting (ec3d_vtk_fields_begin): the field kernel on the
        handle's stream, the copy into one of two pinned host buffers on a side stream.  Returns the slot to hand to
        :meth:`vtk_fields_wait`; the caller goes on with the next step's rhs_step / solve_resident."""
        slot = C.c_int32(0)
        _chk(self.L, self.L.ec3d_vtk_fields_begin(self.h, np.ascontiguousarray(delta, np.float64), int(big_endian),
                                                  C.byref(slot)), "ec3d_vtk_fields_begin")
        return slot.value

    def vtk_fields_wait(self, slot: int, big_endian: bool = True):
        """Block until slot's copy has landed; dict(A, eddy (None without conductors), source, B) of (ncells, 3)
        arrays that VIEW the library's pinned buffer (dtype '>f4' when big_endian): valid until the third
        vtk_fields_begin after the one that returned this slot (VTK_SLOTS buffers, taken in turn)."""
        p = [C.POINTER(C.c_float)() for _ in range(4)]
        n = C.c_int64(0)
        _chk(self.L, self.L.ec3d_vtk_fields_wait(self.h, slot, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]), C.byref(p[3]),
                                                 C.byref(n)), "ec3d_vtk_fields_wait")
        dt = np.dtype(">f4") if big_endian else np.dtype(np.float32)

        def view(q):
            if not q:
                return None
            return np.ctypeslib.as_array(q, shape=(n.value * 3,)).view(dt).reshape(n.value, 3)
        return dict(A=view(p[0]), eddy=view(p[1]), source=view(p[2]), B=view(p[3]))

    def true_residual(self):
        """(||B - A X|| / ||B||, ||B||) of the resident vectors, computed on the device."""
        rel, bn = C.c_double(0), C.c_double(0)
        _chk(self.L, self.L.ec3d_true_residual(self.h, C.byref(rel), C.byref(bn)), "ec3d_true_residual")
        return rel.value, bn.value

    def spmv(self, x):
        y = np.empty(self.n)
        _chk(self.L, self.L.ec3d_spmv(self.h, np.ascontiguousarray(x, np.float64), y), "ec3d_spmv")
        return y

    # ---- measurement ----------------------------------------------------------------------
    def time_kernel(self, name: str, reps: int = 20) -> float:
        ms = C.c_double(0)
        _chk(self.L, self.L.ec3d_time_kernel(self.h, KERNEL[name], reps, C.byref(ms)), "ec3d_time_kernel")
        return ms.value

    def time_iterations(self, iters: int) -> float:
        ms = C.c_double(0)
        _chk(self.L, self.L.ec3d_time_iterations(self.h, iters, C.byref(ms)), "ec3d_time_iterations")
        return ms.value

    def iterate_begin(self):
        _chk(self.L, self.L.ec3d_iterate_begin(self.h), "ec3d_iterate_begin")

    def iterate(self, first_iter: int, count: int, per_kernel: bool = False):
        """Enqueue `count` iterations (exits disabled).  per_kernel=True: synchronises and returns
        the average duration of K1..K5 in ms (hipEvents on the library's stream)."""
        if not per_kernel:
            _chk(self.L, self.L.ec3d_iterate(self.h, first_iter, count, None), "ec3d_iterate")
            return None
        ms = np.zeros(5)
        _chk(self.L, self.L.ec3d_iterate(self.h, first_iter, count, ms.ctypes.data), "ec3d_iterate")
        return dict(zip(("k1", "k2", "k3", "k4", "k5"), ms.tolist()))

    # ---- multi-rank building blocks (see dist.py) -------------------------------------------
    def set_stream(self, stream_ptr: int | None):
        _chk(self.L, self.L.ec3d_set_stream(self.h, C.c_void_p(stream_ptr or 0)), "ec3d_set_stream")

    def vector_layout(self):
        g, n, npad, halo = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _chk(self.L, self.L.ec3d_vector_layout(self.h, C.byref(g), C.byref(n), C.byref(npad), C.byref(halo)),
             "ec3d_vector_layout")
        return dict(ghost=g.value, n=n.value, n_pad=npad.value, halo=halo.value)

    def adopt_vectors(self, device_ptr: int):
        _chk(self.L, self.L.ec3d_adopt_vectors(self.h, C.c_void_p(device_ptr)), "ec3d_adopt_vectors")

    def dist_configure(self, nranks: int, lsum_ptr: int, gsum_ptr: int):
        _chk(self.L, self.L.ec3d_dist_configure(self.h, nranks, C.c_void_p(lsum_ptr), C.c_void_p(gsum_ptr)),
             "ec3d_dist_configure")

    def dist_step(self, stage: int, it: int = 0, tol: float = 0.0):
        _chk(self.L, self.L.ec3d_dist_step(self.h, stage, it, float(tol)), "ec3d_dist_step")

    def dist_set_boundary_rows(self, ranges):
        """ranges: [(lo, hi)] device rows this rank sends in a halo exchange; enables the K2/K5 split stages."""
        lo = np.ascontiguousarray([r[0] for r in ranges] or [0], np.int64)
        hi = np.ascontiguousarray([r[1] for r in ranges] or [0], np.int64)
        on = C.c_int32(0)
        _chk(self.L, self.L.ec3d_dist_set_boundary_rows(self.h, len(ranges), lo, hi, C.byref(on)),
             "ec3d_dist_set_boundary_rows")
        return bool(on.value)

    def can_overlap(self) -> bool:
        return bool(self.L.ec3d_can_overlap(self.h))

    def read_state_async(self, pinned_int32_ptr: int):
        """Enqueue a copy of the stop flag into pinned host memory (see include/ec3d_hip.h)."""
        _chk(self.L, self.L.ec3d_read_state_async(self.h, C.c_void_p(pinned_int32_ptr)), "ec3d_read_state_async")

    def read_state(self):
        si, sk, bn = C.c_int32(0), C.c_int32(0), C.c_double(0)
        _chk(self.L, self.L.ec3d_read_state(self.h, C.byref(si), C.byref(sk), C.byref(bn)), "ec3d_read_state")
        return si.value, sk.value, bn.value

    def fusion(self):
        """(K2 inside K3, K5 inside the next K1) as 0/1: which launches one iteration of this handle is made of."""
        a, b = C.c_int32(0), C.c_int32(0)
        _chk(self.L, self.L.ec3d_get_fusion(self.h, C.byref(a), C.byref(b)), "ec3d_get_fusion")
        return a.value, b.value

    def x_interval(self) -> int:
        """Iterations between two updates of X (ec3d_get_x_interval): 1, or D when K4 defers the update."""
        d = C.c_int32(0)
        _chk(self.L, self.L.ec3d_get_x_interval(self.h, C.byref(d)), "ec3d_get_x_interval")
        return d.value

    def x_groups(self):
        """(how, launches so far): whether the groups of deferred X updates are applied by launches of their own instead of
        by every D-th K4 -- 0: no; 1: on a second stream beside the iteration; 2: on the iteration's own stream, behind the
        K4 of each group's last iteration (ec3d_get_x_groups)."""
        a, b = C.c_int32(0), C.c_int32(0)
        _chk(self.L, self.L.ec3d_get_x_groups(self.h, C.byref(a), C.byref(b)), "ec3d_get_x_groups")
        return a.value, b.value

    def k4_as_spmv(self) -> bool:
        """K4 computes AS = A*S again instead of reading a stored AS (ec3d_get_k4_form)."""
        d = C.c_int32(0)
        _chk(self.L, self.L.ec3d_get_k4_form(self.h, C.byref(d)), "ec3d_get_k4_form")
        return bool(d.value)

    def band_placement(self):
        """(candidate SpMV times in us, index kept) of the placement probe of plain band streams; ([], -1): none ran."""
        us = np.zeros(16)
        tried, kept = C.c_int32(0), C.c_int32(-1)
        _chk(self.L, self.L.ec3d_get_band_placement(self.h, 16, us.ctypes.data, C.byref(tried), C.byref(kept)),
             "ec3d_get_band_placement")
        return [float(v) for v in us[:tried.value]], kept.value

    def vector_placement(self):
        """(candidate iteration times in us, index kept, search time in ms) of the placement probe of the work vectors
        (ec3d_get_vector_placement); ([], -1, 0.0): none ran."""
        us = np.zeros(16)
        tried, kept, ms = C.c_int32(0), C.c_int32(-1), C.c_double(0.0)
        _chk(self.L, self.L.ec3d_get_vector_placement(self.h, 16, us.ctypes.data, C.byref(tried), C.byref(kept), C.byref(ms)),
             "ec3d_get_vector_placement")
        return [float(v) for v in us[:tried.value]], kept.value, ms.value

    def place_vectors(self, candidates: int = 4):
        """Run the placement search of the work vectors now (ec3d_place_vectors): every vector is zero afterwards."""
        _chk(self.L, self.L.ec3d_place_vectors(self.h, candidates), "ec3d_place_vectors")
        return self.vector_placement()

    def restart_count(self) -> int:
        """Times the restart of src/solvers.f90:47-49 fired in the last solve (counted on the device)."""
        k = C.c_int32(0)
        _chk(self.L, self.L.ec3d_get_restart_count(self.h, C.byref(k)), "ec3d_get_restart_count")
        return k.value

    def synchronize(self):
        _chk(self.L, self.L.ec3d_device_synchronize(self.h), "ec3d_device_synchronize")


class _SlabView(EC3DSolver):
    """A slab of an EC3DMulti as an EC3DSolver for introspection (geometry, info, read_state); the multi
    handle owns it."""

    def __init__(self, L, h):
        self.L, self.h = L, h

    def close(self):
        self.h = C.c_void_p()

    __del__ = close


class EC3DMulti:
    """N GPUs behind one handle (include/ec3d_hip.h section 2c): the library cuts the grid into z-slabs, keeps
    one host thread per slab and moves halo planes and partial sums between the devices itself.  Host vectors
    are in the reference's global numbering.  devices=None: GPUs 0 .. nranks-1 (raises "needs N devices" when
    the machine has fewer); a list may name one GPU several times (several slabs on one card)."""

    def __init__(self, nranks: int, devices=None, dictionary: bool | None = None, structured: bool | None = None):
        self.L = load_library()
        self.h = C.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, np.int32)
        if dev is not None and len(dev) != nranks:
            raise ValueError("devices must name one device per rank")
        _chk(self.L, self.L.ec3d_multi_create(C.byref(self.h), int(nranks), None if dev is None else dev.ctypes.data),
             "ec3d_multi_create")
        self.nranks = int(nranks)
        self.rank = None           # (set on the handle of ONE rank of a one-process-per-GPU job: for_rank)
        if dictionary is not None or structured is not None:
            _chk(self.L, self.L.ec3d_multi_set_format(self.h, -1 if dictionary is None else int(bool(dictionary)),
                                                      -1 if structured is None else int(bool(structured))),
                 "ec3d_multi_set_format")

    @staticmethod
    def rccl_unique_id() -> bytes:
        """A fresh RCCL unique id (128 bytes; ec3d_rccl_unique_id): made on ONE rank, handed to all."""
        L = load_library()
        buf = C.create_string_buffer(128)
        _chk(L, L.ec3d_rccl_unique_id(buf), "ec3d_rccl_unique_id")
        return buf.raw

    @classmethod
    def for_rank(cls, rank: int, world: int, device: int, id_halo: bytes, id_sum: bytes, dictionary: bool | None = None,
                 structured: bool | None = None, rehearse=None):
        """One process per GPU (ec3d_multi_create_rank): this process's slab of a `world`-rank job on `device`, RCCL
        between the ranks, the iteration loop enqueued from C++.  Same methods as the one-process handle; host vectors are
        global on every rank (download / solve fill this rank's planes only).  rehearse=(as_rank, as_world): one rank of
        a larger job on one GPU, its neighbours mapped to itself -- for timing."""
        self = cls.__new__(cls)
        self.L = load_library()
        self.h = C.c_void_p()
        ar, aw = rehearse if rehearse else (-1, 0)
        if len(id_halo) != 128 or len(id_sum) != 128:
            raise ValueError("an RCCL unique id is 128 bytes")
        _chk(self.L, self.L.ec3d_multi_create_rank(C.byref(self.h), int(rank), int(world), int(device), id_halo, id_sum,
                                                   int(ar), int(aw)), "ec3d_multi_create_rank")
        self.nranks = 1            # LOCAL slabs (slab(0) is this rank's)
        self.rank, self.world = int(rank), int(world)
        if dictionary is not None or structured is not None:
            _chk(self.L, self.L.ec3d_multi_set_format(self.h, -1 if dictionary is None else int(bool(dictionary)),
                                                      -1 if structured is None else int(bool(structured))),
                 "ec3d_multi_set_format")
        return self

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.ec3d_multi_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def assemble_poisson(self, sdx, sdy, sdz, delta=(0.00333, 0.00333, 0.00333), bnd=-0.95):
        BND = np.full(6, float(bnd)) if np.isscalar(bnd) else np.ascontiguousarray(
            np.asarray(bnd, np.float64).T).reshape(-1)
        _chk(self.L, self.L.ec3d_multi_assemble_poisson(self.h, sdx, sdy, sdz, BND,
                                                        np.ascontiguousarray(delta, np.float64)),
             "ec3d_multi_assemble_poisson")

    def assemble(self, geoPHYS, geoPHYS_C, valPHYS, BND, delta, dt):
        """Same arguments as EC3DSolver.assemble (the GLOBAL tables)."""
        sdz, sdy, sdx = geoPHYS.shape
        vp = np.asarray(valPHYS, np.float64)
        _chk(self.L, self.L.ec3d_multi_assemble(
            self.h, sdx, sdy, sdz, np.ascontiguousarray(geoPHYS, np.int8).reshape(-1),
            np.ascontiguousarray(geoPHYS_C, np.int32).reshape(-1),
            np.ascontiguousarray(vp.T).reshape(-1), vp.shape[0],
            np.ascontiguousarray(np.asarray(BND, np.float64).T).reshape(-1),
            np.ascontiguousarray(delta, np.float64), float(dt)), "ec3d_multi_assemble")

    def set_matrix_csr(self, valA, irow, jcol):
        irow = np.ascontiguousarray(irow, np.int32)
        _chk(self.L, self.L.ec3d_multi_set_matrix_csr(self.h, len(irow) - 1, np.ascontiguousarray(valA, np.float64),
                                                      irow, np.ascontiguousarray(jcol, np.int32)),
             "ec3d_multi_set_matrix_csr")

    @property
    def n(self) -> int:
        n = C.c_int64(0)
        _chk(self.L, self.L.ec3d_multi_size(self.h, C.byref(n)), "ec3d_multi_size")
        return n.value

    def slab(self, rank: int):
        """(EC3DSolver view of rank's slab, k0, k1)."""
        h, k0, k1 = C.c_void_p(), C.c_int32(0), C.c_int32(0)
        _chk(self.L, self.L.ec3d_multi_slab(self.h, rank, C.byref(h), C.byref(k0), C.byref(k1)), "ec3d_multi_slab")
        return _SlabView(self.L, h), k0.value, k1.value

    def upload(self, which: str, a):
        a = np.ascontiguousarray(a, np.float64)
        if a.size != self.n:
            raise ValueError(f"expected {self.n} entries (global numbering)")
        _chk(self.L, self.L.ec3d_multi_upload(self.h, VEC[which], a), "ec3d_multi_upload")

    def download(self, which: str):
        a = np.zeros(self.n)
        _chk(self.L, self.L.ec3d_multi_download(self.h, VEC[which], a), "ec3d_multi_download")
        return a

    def solve(self, b, x0, tolerance, itmax):
        """One reference solve (src/solvers.f90:3-50) on all slabs.  Returns (x, iter)."""
        x = np.array(x0, dtype=np.float64, copy=True)
        it = C.c_int32(0)
        _chk(self.L, self.L.ec3d_multi_solve(self.h, np.ascontiguousarray(b, np.float64), x, float(tolerance),
                                             int(itmax), C.byref(it)), "ec3d_multi_solve")
        return x, it.value

    def solve_resident(self, tolerance, itmax, hist_cap: int = 0):
        """Returns (iter, hist) like EC3DSolver.solve_resident; the history is not kept on slabs (empty)."""
        it = C.c_int32(0)
        _chk(self.L, self.L.ec3d_multi_solve_resident(self.h, float(tolerance), int(itmax), C.byref(it)),
             "ec3d_multi_solve_resident")
        return it.value, np.zeros((0, 2))

    def true_residual(self):
        rel, bn = C.c_double(0), C.c_double(0)
        _chk(self.L, self.L.ec3d_multi_true_residual(self.h, C.byref(rel), C.byref(bn)), "ec3d_multi_true_residual")
        return rel.value, bn.value

    def spmv(self, x):
        y = np.zeros(self.n)
        _chk(self.L, self.L.ec3d_multi_spmv(self.h, np.ascontiguousarray(x, np.float64), y), "ec3d_multi_spmv")
        return y

    def rhs_step(self, src_index, src_value, moving: bool = False):
        idx = np.ascontiguousarray(src_index, np.int32)
        val = np.ascontiguousarray(src_value, np.float64)
        _chk(self.L, self.L.ec3d_multi_rhs_step(self.h, int(bool(moving)), len(idx),
                                                idx if len(idx) else np.zeros(1, np.int32),
                                                val if len(val) else np.zeros(1)), "ec3d_multi_rhs_step")

    def post_update(self):
        _chk(self.L, self.L.ec3d_multi_post_update(self.h), "ec3d_multi_post_update")

    def vtk_fields(self, delta, ncells: int, conducting: bool):
        mk = lambda: np.empty((ncells, 3), np.float32)
        fa, fs, fb = mk(), mk(), mk()
        fe = mk() if conducting else None
        _chk(self.L, self.L.ec3d_multi_vtk_fields(self.h, np.ascontiguousarray(delta, np.float64), fa.ctypes.data,
                                                  fe.ctypes.data if conducting else None, fs.ctypes.data,
                                                  fb.ctypes.data), "ec3d_multi_vtk_fields")
        return dict(A=fa, eddy=fe, source=fs, B=fb)

    def vtk_fields_begin(self, delta, big_endian: bool = True) -> int:
        """EC3DSolver.vtk_fields_begin on every slab (ec3d_multi_vtk_fields_begin): nothing waits; returns the slot."""
        slot = C.c_int32(0)
        _chk(self.L, self.L.ec3d_multi_vtk_fields_begin(self.h, np.ascontiguousarray(delta, np.float64), int(big_endian),
                                                        C.byref(slot)), "ec3d_multi_vtk_fields_begin")
        return slot.value

    def vtk_fields_wait(self, slot: int, big_endian: bool = True):
        """dict(A, eddy (None without conductors), source, B); each value is a LIST with one (cells of the slab, 3)
        view per slab, in z order -- together the cells of field_N.vtk in file order (vtk.field_vtk_pieces writes such
        lists part by part; vtk.join_parts makes one array of them).  A slab that holds no conductor contributes
        zeros to eddy.  The views are valid until the third vtk_fields_begin after the one that returned the slot."""
        dt = np.dtype(">f4") if big_endian else np.dtype(np.float32)
        parts = {k: [] for k in ("A", "eddy", "source", "B")}
        at = 0
        for r in range(self.nranks):
            p = [C.POINTER(C.c_float)() for _ in range(4)]
            c0, n = C.c_int64(0), C.c_int64(0)
            _chk(self.L, self.L.ec3d_multi_vtk_fields_wait(self.h, slot, r, C.byref(p[0]), C.byref(p[1]), C.byref(p[2]),
                                                           C.byref(p[3]), C.byref(c0), C.byref(n)),
                 "ec3d_multi_vtk_fields_wait")
            if r == 0 and getattr(self, "rank", None) is not None:
                at = c0.value      # one process per GPU: this rank's slab starts where the ranks below it end
                self.first_cell = at
            if c0.value != at:
                raise EC3DError("ec3d_multi_vtk_fields_wait: the slabs do not tile the grid")
            at += n.value
            for k, q in zip(("A", "eddy", "source", "B"), p):
                parts[k].append(np.ctypeslib.as_array(q, shape=(n.value * 3,)).view(dt).reshape(n.value, 3)
                                if q else None)
        if all(e is None for e in parts["eddy"]):
            parts["eddy"] = None
        else:
            parts["eddy"] = [e if e is not None else np.zeros(a.shape, dt) for e, a in zip(parts["eddy"], parts["A"])]
        return parts

    def iterate_begin(self):
        _chk(self.L, self.L.ec3d_multi_iterate_begin(self.h), "ec3d_multi_iterate_begin")

    def iterate(self, first_iter: int, count: int, per_kernel: bool = False):
        if not per_kernel:
            _chk(self.L, self.L.ec3d_multi_iterate(self.h, first_iter, count, None), "ec3d_multi_iterate")
            return None
        ms = np.zeros(5)
        _chk(self.L, self.L.ec3d_multi_iterate(self.h, first_iter, count, ms.ctypes.data), "ec3d_multi_iterate")
        return dict(zip(("k1", "k2", "k3", "k4", "k5"), ms.tolist()))

    def synchronize(self):
        _chk(self.L, self.L.ec3d_multi_synchronize(self.h), "ec3d_multi_synchronize")

    def plan(self):
        """(plan, x_every): the schedule the job runs (ec3d_multi_plan): 0 five launches, 1 K1 / K3 interior + boundary,
        2 K2 / K5 boundary first, 3 three launches per iteration; iterations between two X updates."""
        pl, xe = C.c_int32(0), C.c_int32(0)
        _chk(self.L, self.L.ec3d_multi_plan(self.h, C.byref(pl), C.byref(xe)), "ec3d_multi_plan")
        return pl.value, xe.value

    def halo_rows(self, rank: int = 0):
        """(sent, received): rows local slab `rank` moves in ONE halo exchange of a vector (ec3d_multi_halo_rows)."""
        a, b = C.c_int64(0), C.c_int64(0)
        _chk(self.L, self.L.ec3d_multi_halo_rows(self.h, int(rank), C.byref(a), C.byref(b)), "ec3d_multi_halo_rows")
        return a.value, b.value

    def iterate_timed(self, first_iter: int, count: int, rank: int = 0):
        """The instrumented pass with the synchronisation points bracketed too (ec3d_multi_iterate_timed): ({stage: ms},
        {"reduction_points": (ms per iteration, points per iteration), "halo_waits": (...)}) for local slab `rank`."""
        ms = np.zeros(5)
        sm = np.zeros(2)
        sn = np.zeros(2, np.int32)
        _chk(self.L, self.L.ec3d_multi_iterate_timed(self.h, first_iter, count, rank, ms.ctypes.data, sm.ctypes.data,
                                                     sn.ctypes.data), "ec3d_multi_iterate_timed")
        return ({f"k{i + 1}": float(ms[i]) for i in range(5)},
                {"reduction_points": (float(sm[0]), int(sn[0])), "halo_waits": (float(sm[1]), int(sn[1]))})

    def rccl_info(self):
        """(ranks of the communicator as RCCL counts them, RCCL version, file of the library) of a for_rank handle."""
        n, v = C.c_int32(0), C.c_int32(0)
        buf = C.create_string_buffer(512)
        _chk(self.L, self.L.ec3d_multi_rccl_info(self.h, C.byref(n), C.byref(v), buf, 512), "ec3d_multi_rccl_info")
        return n.value, v.value, buf.value.decode()

    def api_calls(self, rank: int) -> float:
        """HIP runtime calls per iteration rank `rank`'s host thread issued in the last iterate()."""
        v = C.c_double(0.0)
        _chk(self.L, self.L.ec3d_multi_api_calls(self.h, rank, C.byref(v)), "ec3d_multi_api_calls")
        return v.value
