#!/usr/bin/env python3
"""In-process sweep of launch knobs (environment variables the library reads when it lays out its sweeps), timing
every kernel INSIDE the iteration.  usage: vec_sweep.py cube512|av3|dia512 'K=V,K=V;K=V;...' (configs split by ;)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
import bench

wl = sys.argv[1]
configs = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
KN = ("EC3D_NBLK_K2", "EC3D_NBLK_K4", "EC3D_NBLK_K5", "EC3D_NBLK_SPMV", "EC3D_XCD_MAP", "EC3D_VEC_DEPTH", "EC3D_NT",
      "EC3D_PATCH", "EC3D_FUSE23", "EC3D_FUSE51", "EC3D_MAP_K2", "EC3D_MAP_K4", "EC3D_MAP_K5", "EC3D_DEPTH_K2", "EC3D_DEPTH_K4", "EC3D_DEPTH_K5")
with E.EC3DSolver(dictionary=not wl.startswith("dia")) as s:
    if wl == "av3":
        geo, geoC, valPHYS, BND, delta, dt, b = bench.av_system(3)
        s.assemble(geo, geoC, valPHYS, BND, delta, dt)
        n = len(b)
    else:
        N = int(wl[-3:])
        s.assemble_poisson(N, N, N)
        n = N ** 3
        b = bench.bar_rhs(N)
    s.upload("B", b)
    for cfg in configs:
        for k in KN:
            os.environ.pop(k, None)
        for kv in filter(None, cfg.split(",")):
            k, v = kv.split("=")
            os.environ["EC3D_" + k] = v
        s.set_workgroups(0)
        s.upload("X", np.zeros(n))
        s.iterate_begin()
        s.iterate(1, 3)
        s.synchronize()
        ms = s.iterate(4, 20, per_kernel=True)
        ms2 = s.iterate(24, 20, per_kernel=True)
        m = {k: min(ms[k], ms2[k]) for k in ms}
        print(f"{wl:8s} {cfg:44s} " + " ".join(f"{k}={1e3 * m[k]:7.1f}" for k in ("k1", "k2", "k3", "k4", "k5")) +
              f" sum={1e3 * sum(m.values()):7.1f} us  wg={s.geometry(2).nblk}/{s.geometry(0).nblk}/{s.geometry(1).nblk}", flush=True)
