#!/usr/bin/env python3
"""On-box probe for BASELINE config 3 at its stated size (ec_src_move_hole resampled to 256^3, n = 53.2 M: bench.py's
`av256` system): per-stage times of the iteration, the bare SpMV, tile census.  Knobs come from the environment
(EC3D_*), one process per setting, so a shell loop compares settings inside one gpurun call.
usage: av256_perf.py [label] [sdx sdy sdz]   (STEM=LIM: the LIM geometry, e.g. 384 192 128 = BASELINE config 5)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import eddy_currents_3d_amd as E

label = sys.argv[1] if len(sys.argv) > 1 else "default"
dims = tuple(int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (256, 256, 256)
model, t, idx, val, moving = bench.av256_system(dims, os.environ.get("STEM", "ec_src_move_hole"))
if os.environ.get("AIR"):      # no conductor: the three A blocks alone (what the format costs without couplings)
    t["geoPHYS_C"] = np.zeros_like(t["geoPHYS_C"])
    t["ncells0"] = 0
with E.EC3DSolver() as s:
    t0 = time.perf_counter()
    s.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
    ta = time.perf_counter() - t0
    n = s.n
    s.upload("X", np.zeros(n))
    s.upload("B", np.zeros(n))
    s.rhs_step(idx, val, moving=moving)
    mi = s.info
    g0, g1 = s.geometry(0), s.geometry(1)
    rows = int(mi.n)
    head = (f"[{label}] grid {dims} n={n} rows={rows} ncells0={int(t['ncells0'])} nblk={g0.nblk}/{g1.nblk} zm_tpp={g1.zm_tpp} "
            f"ulist={g0.ulist_n} fusion={s.fusion()} xint={s.x_interval()} assemble={ta * 1e3:.0f}ms")
    out = []
    for k in ("spmv", "k1", "k3"):
        try:
            out.append(f"{k}={s.time_kernel(k, 30) * 1e3:.1f}us")
        except E.EC3DError:
            out.append(f"{k}=fused")
    s.iterate_begin()
    s.iterate(1, 5)
    s.synchronize()
    t0 = time.perf_counter()
    K = 100
    s.iterate(6, K)
    s.synchronize()
    el = time.perf_counter() - t0
    km = s.iterate(6 + K, 40, per_kernel=True)
    B = {"k1": 25, "k2": 24, "k3": 17, "k4": 56, "k5": 32}
    out.append(f"iter={el / K * 1e6:.1f}us -> {n * K / el / 1e9:.2f} GDOF.it/s; stages " +
               " ".join(f"{k}={v * 1e3:.1f}us({B.get(k, 0) * rows / v / 1e6 / 8000:.2f})" for k, v in km.items()))
    print(head, " ".join(out), flush=True)
