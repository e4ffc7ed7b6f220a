"""Two OS processes, one GPU, torch.distributed (gloo moves the GPU tensors through the host): the real
multi-process path -- HipSlabOps + SlabSolver + process group -- on real kernels.  RCCL itself needs one
GPU per rank and is exercised by the driver's multi-GPU bench; this covers everything around it."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu
N, TOL = 64, 1e-8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    import sys
    sys.path.insert(0, REPO)
    import torch
    import torch.distributed as dist
    from eddy_currents_3d_amd.dist import SlabSolver
    from bench import bar_rhs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        s = SlabSolver.poisson_cube(N, rank, world, device=0)
        s.set_rhs(bar_rhs(N, s.k0, s.k1), np.zeros(s.n_local))
        it = s.solve(TOL, 20000, poll=6)
        x = s.gather_x()
        # the bench's timed path as well
        s.set_rhs(bar_rhs(N, s.k0, s.k1), np.zeros(s.n_local))
        s.iterate_begin()
        s.iterate(1, 5)
        kms = s.iterate(6, 3, per_kernel=True)
        assert set(kms) == {"k1", "k2", "k3", "k4", "k5"}
        if rank == 0:
            np.save(out, np.concatenate([[it], x]))
        s.ops.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu(tmp_path):
    out = str(tmp_path / "x.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = np.load(out)
    it, x = int(res[0]), res[1:]
    import eddy_currents_3d_amd as E
    from bench import bar_rhs
    b = bar_rhs(N)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        xr, itr, _ = s.solve(b, np.zeros(N ** 3), TOL, 20000)
        res_norm = np.linalg.norm(b - s.spmv(x)) / np.linalg.norm(b)
    print(f"2 processes on one GPU: iter {it} / undivided {itr}, true residual {res_norm:.2e}")
    assert res_norm < 5 * TOL
    assert np.linalg.norm(x - xr) <= 1e-5 * np.linalg.norm(xr)
