#!/usr/bin/env python3
"""Does a plain streaming kernel on a second stream find bandwidth that the three launches of the 512^3 iteration leave
unused?  The iteration alone, then beside device-to-device copies of 1 GiB (2 GiB of traffic each) queued on a torch
stream: per-iteration time, copy traffic per iteration, and the total rate of both together.
usage: overlap_probe.py [grid] [copies]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import eddy_currents_3d_amd as E
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ncopy = int(sys.argv[2]) if len(sys.argv) > 2 else 400
n = N ** 3
with E.EC3DSolver() as s:
    s.assemble_poisson(N, N, N)
    s.upload("B", bench.bar_rhs(N))
    s.upload("X", np.zeros(n))
    s.iterate_begin()
    s.iterate(1, 8)
    s.synchronize()
    iters = 96
    alone = min(s.time_iterations(iters), s.time_iterations(iters)) / iters
    bytes_iter = 117.0 * n
    side = torch.cuda.Stream(priority=0)
    a = torch.zeros(1 << 27, dtype=torch.float64, device="cuda")
    b = torch.zeros(1 << 27, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    evs = []
    with torch.cuda.stream(side):
        for i in range(ncopy):
            b.copy_(a, non_blocking=True)
            e = torch.cuda.Event()
            e.record(side)
            evs.append(e)
    t = s.time_iterations(iters) / iters          # ms per iteration while the copies run
    done = sum(1 for e in evs if e.query())
    torch.cuda.synchronize()
    cp = done * 2.0 * (1 << 30) / iters             # copy traffic per iteration
    print(f"{N}^3: iteration alone {1e3 * alone:.1f} us ({bytes_iter / alone / 1e6:.0f} GB/s on 117 B/row); beside "
          f"{done} of {ncopy} copies: {1e3 * t:.1f} us, copy traffic {cp / 1e9:.2f} GB per iteration -> together "
          f"{(bytes_iter + cp) / t / 1e6:.0f} GB/s; the X updates need {20.0 * n / 1e9:.2f} GB per iteration", flush=True)
