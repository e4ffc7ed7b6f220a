#!/usr/bin/env python3
"""Same-box A/B timing: per-kernel averages INSIDE the iteration (hipEvents on the library's stream, as bench.py
measures them) and the bare SpMV, for one workload.  Boxes differ by up to 10 %, so variants are only ever compared
within one gpurun call; EC3D_LIB selects another build of the library, the other knobs are environment variables
read by the library.   usage: ab_perf.py cube512|cube256|dia512|av1|av3|lim|hole [label]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E

wl = sys.argv[1] if len(sys.argv) > 1 else "cube512"
label = sys.argv[2] if len(sys.argv) > 2 else ""
import bench
with E.EC3DSolver(dictionary=not wl.startswith("dia")) as s:
    if wl.startswith("av"):          # av1 = the shipped compare_to_Elmer grid (0.79 M unknowns), av3 = refined x3
        geo, geoC, valPHYS, BND, delta, dt, b = bench.av_system(int(wl[2:]))
        s.assemble(geo, geoC, valPHYS, BND, delta, dt)
        n = len(b)
    elif wl in ("lim", "hole", "lim0", "hole0"):   # BASELINE configs 5 / 3: LIM at 384x192x128, ec_src_move_hole at
        from eddy_currents_3d_amd import vxc          # 256x256x60; lim0 / hole0: the shipped files' own grids
        case, dims = ("LIM", (384, 192, 128)) if wl.startswith("lim") else ("ec_src_move_hole", (256, 256, 60))
        g = np.load(os.path.join(os.path.dirname(bench.__file__), "tests", "golden", f"g4_{case}.npz"))
        model = vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                             tuple(float(x) for x in g["adj"]))
        if not wl.endswith("0"):
            model = vxc.resample(model, *dims)
        t = vxc.domain_tables(model)
        s.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
        n = s.n
        b = np.random.Generator(np.random.PCG64(7)).standard_normal(n)
    elif wl.startswith("box:"):      # box:512x512x128 -- the single-component operator on any grid
        dx, dy, dz = (int(v) for v in wl[4:].split("x"))
        s.assemble_poisson(dx, dy, dz)
        n = dx * dy * dz
        b = np.random.Generator(np.random.PCG64(7)).standard_normal(n)
    else:
        N = int(wl[-3:])
        s.assemble_poisson(N, N, N)
        n = N ** 3
        b = bench.bar_rhs(N)
    s.upload("B", b)
    s.upload("X", np.zeros(n))
    s.iterate_begin()
    s.iterate(1, 5)
    s.synchronize()
    K = 40
    ms = s.iterate(6, K, per_kernel=True)
    ms2 = s.iterate(6 + K, K, per_kernel=True)
    sp = s.time_kernel("spmv", 30)
    tot = sum(ms2.values())
    # the iteration WITHOUT an event at every kernel boundary (two events around 200 iterations): what a solve pays;
    # an event record between two dependent launches costs the stream a few microseconds of its own
    s.time_iterations(20)
    itr = min(s.time_iterations(200), s.time_iterations(200)) / 200
    print(f"{wl:8s} {label:28s} " + " ".join(f"{k}={1e3 * min(ms[k], ms2[k]):7.1f}" for k in ("k1", "k2", "k3", "k4", "k5")) +
          f" sum={1e3 * tot:7.1f} iter={1e3 * itr:7.1f} spmv={1e3 * sp:7.1f} us  wg={s.geometry(0).nblk}/{s.geometry(1).nblk}"
          + (f"  search {s.vector_placement()[2]:.0f} ms, kept {s.vector_placement()[1]} of {len(s.vector_placement()[0])}"
             if hasattr(s, "vector_placement") and s.vector_placement()[0] else ""), flush=True)
