#!/usr/bin/env python3
"""Scratch sweep: producers whose output stays cacheable (EC3D_KEEP bits: 1 AP, 2 S, 8 R, 32 P) against the size of the
vectors.  Cubes with whole-tile planes, per-iteration time (sum of the per-kernel averages inside the iteration)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
dims = [(int(a.split("x")[0]), int(a.split("x")[1]), int(a.split("x")[2])) for a in sys.argv[1].split(",")]
masks = [int(m) for m in sys.argv[2].split(",")]
for (sx, sy, sz) in dims:
    n = sx * sy * sz
    row = []
    for m in masks:
        os.environ.pop("EC3D_NT", None)
        os.environ.pop("EC3D_KEEP", None)
        if m == -1:
            os.environ["EC3D_NT"] = "0"     # no nontemporal streams at all
        elif m != -2:                       # -2: the library's own policy
            os.environ["EC3D_KEEP"] = str(m)
        with E.EC3DSolver() as s:
            s.assemble_poisson(sx, sy, sz)
            s.upload("B", np.ones(n)); s.upload("X", np.zeros(n))
            s.iterate_begin(); s.iterate(1, 5); s.synchronize()
            a = s.iterate(6, 40, per_kernel=True); b = s.iterate(46, 40, per_kernel=True)
            t = sum(min(a[k], b[k]) for k in a) * 1e3
        row.append(t)
    base = row[0]
    print(f"{sx}x{sy}x{sz} n={n/1e6:6.1f}M vec={n*8/2**20:6.0f}MiB  " + "  ".join(f"m{m}={t:7.1f}({100*(t/base-1):+5.1f}%)" for m, t in zip(masks, row)), flush=True)
