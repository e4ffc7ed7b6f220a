"""Build libec3d_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m eddy_currents_3d_amd.build [--force]

-ffp-contract=off is part of the numerical contract, not a tuning flag: the reference object code
has no fused multiply-add and the kernels are bandwidth-bound anyway.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libec3d_hip.so")
SOURCES = ["ec3d_kernels.hip", "ec3d_context.hip", "ec3d_solve.hip", "ec3d_measure.hip", "ec3d_dist.hip", "ec3d_multi.hip", "ec3d_dropin.hip", "ec3d_assemble.hip", "ec3d_rhs.hip", "ec3d_output.hip", "ec3d_format.cpp", "ec3d_sav_csr.cpp"]
HEADERS = [os.path.join(CSRC, "ec3d_internal.hpp"), os.path.join(os.path.dirname(PKG), "include", "ec3d_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.sep not in p or os.path.exists(p)):
            return p
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not stale():
        return LIB
    cmd = [hipcc(), *FLAGS, *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
