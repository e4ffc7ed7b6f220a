#!/usr/bin/env python3
"""Probe: does this RCCL let two ranks share ONE device?  (Upstream NCCL refuses: "Duplicate GPU detected".)  Two processes
under torch.distributed.run (gloo group for the id broadcast), both on device 0, ec3d_multi_create_rank with nranks = 2.
If it works, a real two-rank job of the C++/RCCL driver can be tested on a one-GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from eddy_currents_3d_amd.dist import rccl_rank
try:
    m = rccl_rank(rank, world, 0)
except Exception as e:
    print(f"rank {rank}: ec3d_multi_create_rank failed: {e}", flush=True)
    dist.barrier()
    sys.exit(3)
N = 64
m.assemble_poisson(N, N, N)
b = np.random.Generator(np.random.PCG64(1)).standard_normal(N ** 3)
x, it = m.solve(b, np.zeros(N ** 3), 1e-8, 5000)
print(f"rank {rank}: two ranks on one device: iter {it}, plan {m.plan()}", flush=True)
m.close()
dist.barrier()
