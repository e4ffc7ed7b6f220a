"""Build libec3d_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m eddy_currents_3d_amd.build [--force]

-ffp-contract=off is part of the numerical contract, not a tuning flag: the reference object code
has no fused multiply-add and the kernels are bandwidth-bound anyway.

Every source is compiled to an object of its own (csrc/build/, git- and gpurun-ignored), stale ones only and side
by side, then linked: a change to one file costs that file's compile time, not the sum.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libec3d_hip.so")
SOURCES = ["ec3d_kernels.hip", "ec3d_context.hip", "ec3d_solve.hip", "ec3d_measure.hip", "ec3d_dist.hip", "ec3d_multi.hip",
           "ec3d_rccl.cpp", "ec3d_dropin.hip", "ec3d_assemble.hip", "ec3d_rhs.hip", "ec3d_output.hip", "ec3d_format.cpp",
           "ec3d_sav_csr.cpp"]
HEADERS = [os.path.join(CSRC, "ec3d_internal.hpp"), os.path.join(CSRC, "ec3d_rccl.hpp"),
           os.path.join(os.path.dirname(PKG), "include", "ec3d_hip.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


def hipcc() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    raise RuntimeError("hipcc not found")


def _obj(src: str) -> str:
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale_obj(src: str) -> bool:
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in [os.path.join(CSRC, src)] + HEADERS)


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    todo = [s for s in SOURCES if force or _stale_obj(s)]

    def compile_one(src):
        cmd = [cc, *CFLAGS, "-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [cc, *LDFLAGS, *[_obj(s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


# ---- test support (NOT part of the product library) -----------------------------------------------------------
# tests/libec3d_loopback.so: the loopback stand-in for librccl (tests/support/rccl_loopback.cpp) the rank-driver tests
# name in EC3D_RCCL_LIB.  Built here because it needs the same compiler; lives under tests/ and travels to the GPU box
# like the product library (git-ignored, not gpurun-ignored).
LOOPBACK_SRC = os.path.join(os.path.dirname(PKG), "tests", "support", "rccl_loopback.cpp")
LOOPBACK_LIB = os.path.join(os.path.dirname(PKG), "tests", "libec3d_loopback.so")


def build_test_support(force: bool = False, verbose: bool = False) -> str:
    if not os.path.exists(LOOPBACK_SRC):
        return ""
    if not force and os.path.exists(LOOPBACK_LIB) and os.path.getmtime(LOOPBACK_LIB) >= os.path.getmtime(LOOPBACK_SRC):
        return LOOPBACK_LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", LOOPBACK_SRC, "-o", LOOPBACK_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LOOPBACK_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_test_support(force="--force" in sys.argv, verbose=True))
