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


def _av_worker(rank, world, port, out, name="g3_moving_coil_18x16x12", moving=True):
    """The whole resident time loop of the A-V system on two ranks: rhs_step (X halo exchange, global source
    ids) -> solve -> post_update, fields never leaving the GPU between steps."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import load_golden
    from test_gpu_timeloop import coil_sources
    from eddy_currents_3d_amd.dist import HipAVSlabOps, SlabSolver, slab_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        g = load_golden(name)
        n = len(g["irow"]) - 1
        tol, itmax = float(g["tol"]), int(g["itmax"])
        k0, k1 = slab_bounds(g["geoPHYS"].shape[0], rank, world)
        ops = HipAVSlabOps(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]),
                           k0, k1, world)
        assert ops.structured
        s = SlabSolver(ops, rank, world, k0, k1)
        print(f"rank {rank}: planes [{k0}, {k1}), vector kernels split: {s.split_ok}", flush=True)
        ops.set_vector_global("X", np.zeros(n))
        ops.set_vector_global("B", np.zeros(n))
        res = []
        for k in range(len(g["iters"])):
            idx, val = coil_sources(g, k, moving)
            s.rhs_step(idx, val, moving=moving)
            it = s.solve(tol, itmax, poll=4)
            x = np.zeros(n)
            ops.export_owned("X", x)
            xt = torch.from_numpy(x)
            dist.all_reduce(xt)
            res.append(np.concatenate([[it], xt.numpy()]))
            s.post_update()
        if rank == 0:
            np.save(out, np.stack(res))
        ops.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,name,moving", [(2, "g3_moving_coil_18x16x12", True), (3, "g3_moving_coil_18x16x12", True),
                                               (4, "g2v_conducting_moving_16x15x14", False)])
def test_ranks_run_the_av_time_loop(tmp_path, world, name, moving):
    """Several ranks, thin slabs: with 4 ranks on 14 planes the inner ranks are all boundary (they run the whole
    kernels in the shared exchange order) while the outer ranks split theirs -- mixed capabilities over real
    processes."""
    from conftest import load_golden
    out = str(tmp_path / "av.npy")
    mp.spawn(_av_worker, args=(world, _free_port(), out, name, moving), nprocs=world, join=True)
    res = np.load(out)
    g = load_golden(name)
    tol = float(g["tol"])
    for k, it_ref in enumerate(g["iters"]):
        it, x = int(res[k, 0]), res[k, 1:]
        xr = g[f"xout{k}"]
        rel = np.linalg.norm(x - xr) / np.linalg.norm(xr)
        print(f"step {k}: iter {world} ranks {it} / reference {int(it_ref)}, rel diff {rel:.2e}")
        assert rel <= 10 * tol


def _host_worker(rank, world, port, out):
    """host.run_slabs: the reference's whole run of a model (sources, motion, time loop, output) on two ranks."""
    import sys
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import load_golden
    from eddy_currents_3d_amd import host, vxc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        g = load_golden("g4_LIM")
        model = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                             tuple(float(x) for x in g["adj"]))
        log = host.run_slabs(model, rank, world, device=0, steps=3, out_dir=out)
        # the same with the fields gathered on rank 0 and written inside the loop, as rounds 1-3 did
        log2 = host.run_slabs(model, rank, world, device=0, steps=3, out_dir=os.path.join(out, "gathered"),
                              overlap_output=False)
        if rank == 0:
            assert [i["iter"] for i in log] == [i["iter"] for i in log2]
            np.save(os.path.join(out, "iters.npy"), np.array([i["iter"] for i in log]))
    finally:
        dist.destroy_process_group()


def test_two_ranks_run_a_shipped_model_end_to_end(tmp_path):
    """LIM.vxc (12 coil materials, three-phase currents, moving primary, anisotropic grid) on two ranks: the
    reference's iteration counts, and output files equal to the single-GPU run's up to the solver tolerance."""
    from conftest import load_golden
    out = str(tmp_path)
    mp.spawn(_host_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    g = load_golden("g4_LIM")
    iters = np.load(os.path.join(out, "iters.npy"))
    print("2 ranks:", iters, "reference:", g["iters"])
    assert np.all(np.abs(iters - g["iters"]) <= np.maximum(3, 0.15 * g["iters"]))
    names = ["field_1.vtk", "field_2.vtk", "src_1.vtk", "src_2.vtk"]
    assert sorted(f for f in os.listdir(out) if f.endswith(".vtk")) == names
    # every rank put its own cells into the files, beside the next step's solve: the bytes of the gathered path
    import filecmp
    match, mismatch, errors = filecmp.cmpfiles(out, os.path.join(out, "gathered"), names, shallow=False)
    assert sorted(match) == names and not mismatch and not errors
