"""N > 1 path on CPU: the slab schedule of eddy_currents_3d_amd/dist.py (halo send/recv, all_gather of
the partial sums, rank-ordered reduction, stop flag agreement) over gloo with world_size 2, 3 and 8,
with a numpy stand-in for the per-slab device ops.  Checked against the oracle's serial solve."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import REPO

N = 12
TOL = 1e-8


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out, producer_side=False, N=N):
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch.distributed as dist
    from oracle import oracle as O
    from eddy_currents_3d_amd.dist import SlabSolver, slab_bounds
    from slab_numpy_ops import NumpySlabOps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k0, k1 = slab_bounds(N, rank, world)
        ops = NumpySlabOps(O, N, N, N, k0, k1, world)
        ops.producer_side_overlap = producer_side
        if producer_side:
            ops.overlap = False
        s = SlabSolver(ops, rank, world, k0, k1)
        if producer_side:
            from eddy_currents_3d_amd.dist import ITER_PLAN_VSPLIT
            assert s.iter_plan is ITER_PLAN_VSPLIT
        b = O.bar_rhs(N).reshape(N, N * N)[k0:k1].reshape(-1)
        s.set_rhs(b, np.zeros(s.n_local))
        it = s.solve(TOL, 10000, poll=4)
        x = s.gather_x()
        if rank == 0:
            np.save(out, np.concatenate([[it], x]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("producer_side", [False, True])
def test_slab_solver_over_gloo_matches_serial_oracle(oracle, tmp_path, world, producer_side):
    """producer_side: the A-V slabs' schedule (K2/K5 boundary rows first, asynchronous send/recv started
    behind them and joined before the consumer) over real processes and real non-blocking transfers."""
    out = str(tmp_path / "x.npy")
    N = 32 if world == 8 else 12   # 8 ranks: 4 planes each, so the split (interior + boundary) schedules run
    mp.spawn(_worker, args=(world, _free_port(), out, producer_side, N), nprocs=world, join=True)
    res = np.load(out)
    it, x = int(res[0]), res[1:]
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    b = oracle.bar_rhs(N)
    xo, ito, _, _ = oracle.bicgstab_wr(valA, irow, jcol, b, np.zeros(N ** 3), TOL, 10000)
    res_norm = np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, x)) / np.linalg.norm(b)
    assert res_norm < 5 * TOL
    assert np.linalg.norm(x - xo) <= 1e-5 * np.linalg.norm(xo)
    assert abs(it - ito) <= 0.35 * ito


def test_slab_bounds_partition():
    from eddy_currents_3d_amd.dist import slab_bounds
    for sdz in (7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            if world > sdz:
                continue
            b = [slab_bounds(sdz, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == sdz
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [k1 - k0 for k0, k1 in b]
            assert max(sizes) - min(sizes) <= 1


def test_in_process_slabs_equal_single_slab(oracle):
    """Three slabs in one process (same schedule, tensor-to-tensor halos) vs one slab vs the oracle."""
    from eddy_currents_3d_amd.dist import InProcessSlabs, slab_bounds
    from slab_numpy_ops import NumpySlabOps
    b = oracle.bar_rhs(N)
    xs, its = [], []
    for world in (1, 3):
        ops = []
        for r in range(world):
            k0, k1 = slab_bounds(N, r, world)
            o = NumpySlabOps(oracle, N, N, N, k0, k1, world)
            o.set_vector("B", b.reshape(N, N * N)[k0:k1].reshape(-1))
            ops.append(o)
        drv = InProcessSlabs(ops)
        its.append(drv.solve(TOL, 10000))
        xs.append(drv.x())
    assert np.linalg.norm(xs[0] - xs[1]) <= 1e-6 * np.linalg.norm(xs[0])
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    assert np.linalg.norm(b - oracle.spmv_csr(valA, irow, jcol, xs[1])) / np.linalg.norm(b) < 5 * TOL


def _ids_worker(rank, world, port, out):
    import sys
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from eddy_currents_3d_amd.dist import share_unique_ids, rccl_rank
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.Generator(np.random.PCG64(99))       # what rank 0 "made"; the others hold nothing
        ids = [rng.bytes(128), rng.bytes(128)] if rank == 0 else [None, None]
        got = share_unique_ids(ids, rank, world, 0)
        np.save(f"{out}.{rank}.npy", np.frombuffer(got[0] + got[1], np.uint8))
        try:    # a job of another size than the process group's has nobody to carry the ids
            rccl_rank(rank, world + 1, 0)
            refused = False
        except RuntimeError as e:
            refused = "needs torch.distributed" in str(e)
        assert refused
    finally:
        dist.destroy_process_group()


def test_rank_zero_ids_reach_every_rank(tmp_path):
    """What carries the two RCCL unique ids of the one-process-per-GPU driver (dist.rccl_rank -> ec3d_multi_create_rank)
    from rank 0 to the others: one broadcast of 256 bytes on the job's own process group.  gloo, 3 ranks."""
    world, out = 3, str(tmp_path / "ids")
    mp.spawn(_ids_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    rng = np.random.Generator(np.random.PCG64(99))
    want = np.frombuffer(rng.bytes(128) + rng.bytes(128), np.uint8)
    for r in range(world):
        assert np.array_equal(np.load(f"{out}.{r}.npy"), want)
