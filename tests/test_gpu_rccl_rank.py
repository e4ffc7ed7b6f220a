"""One process per GPU, driven from C++ over RCCL (include/ec3d_hip.h section 2c: ec3d_multi_create_rank; what bench.py
runs under torch.distributed.run).  The plans, stages and kernels are those of the one-process handle, which
tests/test_gpu_slab_plans.py and tests/test_gpu_multi.py pin bit for bit; what is specific to this driver is the
transport -- ncclSend / ncclRecv groups for the halo planes, ncclAllGather for the sums -- and the gathered copy of the
sums the kernels then read.  On the one-GPU test box:

* a ONE-rank job (communicators of one rank, the all-gather included) must reproduce the plain handle bit for bit;
* a REHEARSAL of a middle rank of a larger job (its slab, plan and RCCL calls, every neighbour mapped to the process
  itself) must run every plan to the end, with the send / recv groups and all-gathers really issued;
* two ranks in two processes need two GPUs (RCCL refuses two ranks on one device): skipped here, run where there are."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    import eddy_currents_3d_amd as E
    E.load_library()
    return E


def test_one_rank_job_over_rccl_equals_the_plain_handle(E, oracle):
    from eddy_currents_3d_amd.dist import rccl_rank
    N, tol = 24, 1e-8
    b = oracle.bar_rhs(N)
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        x_ref, it_ref, _ = s.solve(b, np.zeros(N ** 3), tol, 10000)
    with rccl_rank(0, 1, 0) as m:
        m.assemble_poisson(N, N, N)
        assert m.n == N ** 3
        x, it = m.solve(b, np.zeros(N ** 3), tol, 10000)
        rel, bn = m.true_residual()
        xs = np.random.Generator(np.random.PCG64(3)).standard_normal(N ** 3)
        y = m.spmv(xs)
    valA, irow, jcol = oracle.poisson_csr(N, N, N)
    assert it == it_ref and np.array_equal(x, x_ref)
    assert bn == pytest.approx(float(np.linalg.norm(b)), rel=1e-13) and rel < 5 * tol
    assert np.array_equal(y, oracle.spmv_csr(valA, irow, jcol, xs))


@pytest.mark.parametrize("dims, knobs, plan", [((128, 8, 96), dict(SLAB_PLAN=1), 1), ((24, 24, 24), {}, 0), ((128, 8, 96), dict(SLAB_PLAN=5), 5),
                                               ((128, 8, 96), dict(FUSE23=2, FUSE51=2, K4S=2, XDEFER=4), 4),
                                               ((128, 8, 96), dict(FUSE23=2, FUSE51=2, K4S=2, XDEFER=4, SLAB_FSPLIT=0), 3)],
                         ids=["interior+boundary", "plain", "both-split", "three-launches-split", "three-launches"])
def test_rehearsal_of_a_middle_rank_runs_every_plan(E, monkeypatch, dims, knobs, plan):
    """Rank 1 of 4 alone on this GPU: both neighbours exist (and are this process), so every halo exchange is a group of
    two sends and two receives on the side stream and every reduction point an all-gather.  Exits disabled; the values
    mean nothing (the slab is wrapped onto itself), the schedule is the real rank's."""
    from eddy_currents_3d_amd.dist import rccl_rank
    for k in ("EC3D_FUSE23", "EC3D_FUSE51", "EC3D_K4S", "EC3D_XDEFER", "EC3D_SLAB_FUSE", "EC3D_SLAB_XDEFER", "EC3D_NT", "EC3D_SLAB_FSPLIT",
              "EC3D_SLAB_PLAN"):
        monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
        monkeypatch.setenv("EC3D_" + k, str(v))
    sdx, sdy, sdz = dims
    n = sdx * sdy * sdz
    with rccl_rank(0, 1, 0, rehearse=(1, 4)) as m:
        m.assemble_poisson(sdx, sdy, sdz)
        view, k0, k1 = m.slab(0)
        assert (k0, k1) == (sdz // 4, sdz // 2)
        assert m.plan()[0] == plan
        m.upload("B", np.random.Generator(np.random.PCG64(1)).standard_normal(n))
        m.upload("X", np.zeros(n))
        m.iterate_begin()
        m.iterate(1, 40)
        m.synchronize()
        km = m.iterate(41, 10, per_kernel=True)
        calls = m.api_calls(0)
    print(f"plan {plan}: {calls:.0f} runtime calls per iteration (launches, events, RCCL), stages "
          + " ".join(f"{k}={v * 1e3:.0f}us" for k, v in km.items()))
    assert calls > 10


@pytest.mark.parametrize("name", ["g2_conducting_hole_16x15x14", "g3_moving_coil_18x16x12"])
def test_one_rank_av_job_over_rccl_reproduces_the_reference_captures(E, name):
    """The full A-V system [Ax | Ay | Az | U] (src/EC3D.f90:408) through the rank handle: native assembly, the reference's
    captured time steps (tests/golden/g2_*, g3_*: b, x_in, x_out, iter of the unmodified program): same iteration counts, x
    within 10*tol of the reference's, and bit-identical to the plain handle."""
    from conftest import load_golden
    from eddy_currents_3d_amd.dist import rccl_rank
    g = load_golden(name)
    tol, itmax = float(g["tol"]), int(g["itmax"])
    with E.EC3DSolver() as s, rccl_rank(0, 1, 0) as m:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        assert m.n == len(g["irow"]) - 1
        for k in (0, 1):
            xs, its, _ = s.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            x, it = m.solve(g[f"b{k}"], g[f"xin{k}"], tol, itmax)
            xr = g[f"xout{k}"]
            assert it == its == int(g["iters"][k]) and np.array_equal(x, xs)
            assert np.linalg.norm(x - xr) <= 10 * tol * np.linalg.norm(xr)


def test_rehearsal_of_a_middle_av_rank(E, monkeypatch):
    """Rank 1 of 3 of an A-V job alone on this GPU (plan 2: K2 / K5 boundary tiles first; four blocks exchanged per
    neighbour, the U block two planes deep), X every fourth iteration forced: the schedule runs to the end."""
    from conftest import load_golden
    from eddy_currents_3d_amd.dist import rccl_rank
    monkeypatch.setenv("EC3D_XDEFER", "4")
    g = load_golden("g2_conducting_hole_16x15x14")
    n = len(g["irow"]) - 1
    with rccl_rank(0, 1, 0, rehearse=(1, 3)) as m:
        m.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        assert m.plan() == (2, 4)
        m.upload("B", g["b0"])
        m.upload("X", np.zeros(n))
        m.iterate_begin()
        m.iterate(1, 30)
        m.synchronize()
        assert m.api_calls(0) > 10


def test_two_ranks_in_two_processes():
    """The real thing: two processes, two GPUs, torch.distributed.run; bench.py verifies A*x over the ranks against one
    GPU bit for bit and the reductions through ||b|| before it times anything, and fails without a number otherwise."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL does not put two ranks on one device)")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(REPO, "bench.py"), "--gpus", "2",
                          "--grid", "256", "--steps", "20", "--warmup", "3"], capture_output=True, text=True, env=env,
                         timeout=500)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and "bit for bit" in line["verified"]
