#!/usr/bin/env python3
"""Per-rank iteration time at the slab shapes of the 512^3 cube, measured on ONE card.

    python tools/slab_shapes.py [iters]

For G = 2, 4, 8 GPUs a rank of the 512^3 job holds 512 x 512 x (512 / G) cells.  Two such slabs on one card (EC3DMulti(2,
devices=[0, 0]): the plan, kernels, exchanges and reduction points of the real job, with local copies for xGMI ones) run
`iters` iterations with exits disabled; the card executes both slabs' launches, so wall time per iteration / 2 is what one
rank's kernels cost (plus what two host threads on one device's runtime add: an upper bound).  Printed per shape: the plan
the library picked, X interval, ms per iteration for the pair and per rank, HIP calls per iteration and rank.  The
undivided 512^3 handle runs in the same process for the ratio."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch  # noqa: F401  (the library then shares torch's HIP runtime)
import eddy_currents_3d_amd as E
from bench import bar_rhs

N = 512
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def timed(s, n):
    s.upload("X", np.zeros(n))
    s.iterate_begin()
    s.iterate(1, 5)
    s.synchronize()
    t0 = time.perf_counter()
    s.iterate(6, iters)
    s.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


with E.EC3DSolver() as s:
    s.assemble_poisson(N, N, N)
    s.upload("B", bar_rhs(N))
    one = timed(s, N ** 3)
    print(f"undivided 512^3: {one:.4f} ms per iteration, fusion {s.fusion()}, X every {s.x_interval()}", flush=True)
for G in (2, 4, 8):
    planes = N // G
    sdz = 2 * planes
    n = N * N * sdz
    rng = np.random.Generator(np.random.PCG64(G))
    with E.EC3DMulti(2, devices=[0, 0]) as m:
        m.assemble_poisson(N, N, sdz)
        m.upload("B", rng.standard_normal(n))
        pair = timed(m, n)
        plan, xd = m.plan()
        calls = [m.api_calls(r) for r in range(2)]
        km = m.iterate(6 + iters, 20, per_kernel=True)
    print(f"G={G}: two slabs of 512x512x{planes} ({N * N * planes / 2**20:.0f} Mi rows each) on one card: plan {plan}, X every {xd}, "
          f"{pair:.4f} ms per iteration for the pair = {pair / 2:.4f} ms per rank; HIP calls per iteration {calls}; "
          f"ratio to the undivided handle {one / (pair / 2):.2f}; rank 0's stages (both slabs share the card) "
          + " ".join(f"{k}={v * 1e3:.0f}us" for k, v in km.items()), flush=True)
