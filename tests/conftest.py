import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/): C restatement of the reference, compiled on first use."""
    from oracle import oracle as O
    O.build(with_ref=False)
    return O


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(params=["auto", "pitched"])
def plane_pitch(request, monkeypatch):
    """Structured A-V form with the default plane pitch (tile-aligned planes only when cheap: never on the
    small fixture grids) and with tile-aligned planes forced (device layout != reference layout, the
    z-marching SpMV map with a ragged column count)."""
    if request.param == "pitched":
        monkeypatch.setenv("EC3D_PITCH", "2")
    else:
        monkeypatch.delenv("EC3D_PITCH", raising=False)
    return request.param


@pytest.fixture(params=["linear", "patch", "patch-fused"])
def sav_tiles(request, monkeypatch):
    """Tiles of the z-marching structured A-V kernels: 512 consecutive cells (sav_pair_zm), runtime-shaped 2-D
    patches (sav_patch_step, forced on the small fixture grids, where the library would not pick them), and the
    patches with K2 inside K3 and K5 inside the next K1 (three launches per iteration)."""
    monkeypatch.setenv("EC3D_SAV_PATCH", "0" if request.param == "linear" else "2")
    fuse = "2" if request.param == "patch-fused" else "0"
    monkeypatch.setenv("EC3D_FUSE23", fuse)
    monkeypatch.setenv("EC3D_FUSE51", fuse)
    return request.param
