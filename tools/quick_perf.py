#!/usr/bin/env python3
"""Quick on-box probe: per-kernel time and algorithmic GB/s on synthetic cubes.
usage: quick_perf.py N [nblk,nblk,...] [dict=0|1] [nt=-1|0|1]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
from eddy_currents_3d_amd.solver import KERNEL_BYTES_PER_ROW

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nblks = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for nblk in nblks:
    with E.EC3DSolver() as s:
        if nblk: s.set_workgroups(nblk)
        t = time.time(); s.assemble_poisson(N, N, N); ta = time.time() - t
        n = N ** 3
        rng = np.random.Generator(np.random.PCG64(1))
        s.upload("B", rng.standard_normal(n)); s.upload("X", np.zeros(n))
        out = [f"N={N} nblk={s.geometry().nblk} dict={s.info.dict_classes}"]
        for k in ("spmv", "k1", "k2", "k3", "k4", "k5"):
            ms = s.time_kernel(k, 20)
            out.append(f"{k}={ms:.3f}ms")
        s.time_iterations(3)
        ms = s.time_iterations(20)
        out.append(f"iter={ms/20:.3f}ms -> {n*20/ms/1e6:.2f} GDOF.it/s ({264*n*20/ms/1e6:.0f} GB/s alg)")
        print(" ".join(out), flush=True)
