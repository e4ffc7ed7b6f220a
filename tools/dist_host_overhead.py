#!/usr/bin/env python3
"""Host-side cost of one distributed iteration (python + ctypes + torch.distributed/RCCL call overhead),
measured on ONE GPU: a 1-rank RCCL group where the halo send/recv pairs go to the rank itself, on a grid so
small that the kernels take no time.  What it prints is the floor of ms/iteration the N-GPU schedule can
reach before the GPU work matters.  usage: dist_host_overhead.py [N=64] [iters=300]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from eddy_currents_3d_amd import dist as D

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = D.SlabSolver.poisson_cube(N, 0, 1, device=0)
ops = s.ops
s.set_rhs(np.random.default_rng(0).standard_normal(s.n_local), np.zeros(s.n_local))
s.iterate_begin()


def p2p(name):
    lo_s, lo_r, hi_s, hi_r = ops.halo_views(name)
    return [dist.P2POp(dist.isend, lo_s, 0), dist.P2POp(dist.irecv, hi_r, 0),
            dist.P2POp(dist.isend, hi_s, 0), dist.P2POp(dist.irecv, lo_r, 0)]


lists = {"P": p2p("P"), "S": p2p("S")}
overlap = ops.can_overlap()
plan = D.ITER_PLAN_OVERLAP if overlap else D.ITER_PLAN


def iteration(it, with_p2p):
    pend = None
    for op in plan:
        if op[0] in ("halo", "halo_start"):
            if with_p2p:
                pend = dist.batch_isend_irecv(lists[op[1]])
                if op[0] == "halo":
                    for r in pend:
                        r.wait()
        elif op[0] == "halo_wait":
            if with_p2p:
                for r in pend:
                    r.wait()
        elif op[0] == "gather":
            s.gather()
        else:
            ops.step(op[1], it, 0.0)


for with_p2p in (False, True):
    with ops.context():
        for it in range(1, 21):
            iteration(it, with_p2p)
        ops.synchronize()
        t0 = time.perf_counter()
        for it in range(21, 21 + iters):
            iteration(it, with_p2p)
        t_enq = time.perf_counter() - t0
        ops.synchronize()
        t_all = time.perf_counter() - t0
    print(f"N={N} overlap_plan={overlap} p2p={'self send/recv' if with_p2p else 'none'}: host enqueue "
          f"{1e3 * t_enq / iters:.3f} ms/iter, enqueue+drain {1e3 * t_all / iters:.3f} ms/iter", flush=True)
ops.close()
dist.destroy_process_group()
