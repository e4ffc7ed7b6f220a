#!/usr/bin/env python3
"""Host-side floor of the in-library multi-GPU loop (csrc/ec3d_multi.hip), measured on ONE GPU.

N slabs of a small grid all sit on device 0, so the kernels are short and what is timed is what the N host threads
need to enqueue one iteration each: 11 launches (5 stages, 2 of them split, 4 partial-sum collapses), 2 halo pulls
(event record, cross-stream waits, 1-2 copies, event record, wait) and 3 reduction points (event record, N-1
cross-stream waits).  Compare with the ~0.47 ms of kernel time a rank has per iteration at 512^3 on 8 GPUs and
~0.06 ms at 256^3.  Also prints the same for a grid large enough that the GPU is the limit (one card shared by N
slabs: only the trend matters).

    python tools/multi_host_overhead.py [--planes-per-slab 16] [--iters 400]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes-per-slab", type=int, default=16)
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--edge", type=int, default=64, help="xy edge of the grid (64: 4096-cell planes, 8 tiles)")
    args = ap.parse_args()
    import eddy_currents_3d_amd as E
    print(f"grid {args.edge}x{args.edge}x(N*{args.planes_per_slab}), {args.iters} iterations, all slabs on device 0")
    for n in (1, 2, 4, 8):
        sdz = n * args.planes_per_slab
        with E.EC3DMulti(n, devices=[0] * n) as m:
            m.assemble_poisson(args.edge, args.edge, sdz)
            m.upload("B", np.random.Generator(np.random.PCG64(1)).standard_normal(m.n))
            m.upload("X", np.zeros(m.n))
            m.iterate_begin()
            m.iterate(1, 50)
            m.synchronize()
            t0 = time.perf_counter()
            m.iterate(51, args.iters)          # returns when every thread has ENQUEUED its iterations
            t1 = time.perf_counter()
            m.synchronize()
            t2 = time.perf_counter()
            plan = "overlap" if m.slab(0)[0].can_overlap() else "plain"
            calls = [m.api_calls(r) for r in range(n)]
        print(f"N={n}: enqueue {1e6 * (t1 - t0) / args.iters:7.1f} us/iteration per rank thread (all threads in parallel), "
              f"until drained {1e6 * (t2 - t0) / args.iters:7.1f} us/iteration, plan {plan}, rows per slab "
              f"{args.edge * args.edge * args.planes_per_slab}; HIP runtime calls per iteration and rank: "
              f"{' '.join(f'{c:.0f}' for c in calls)}")


if __name__ == "__main__":
    main()
