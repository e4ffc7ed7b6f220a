#!/usr/bin/env python3
"""One rank of the 512^3 job on G GPUs, alone on ONE card, through the driver the launcher's form of bench.py uses: the
C++ iteration loop over RCCL (ec3d_multi_create_rank with as_world = G: that rank's slab, plan, launches, send / recv
groups and all-gathers, its neighbours mapped to this process).  Prints, per G, ms per iteration of the rank, the host
thread's enqueue time per iteration and its runtime calls per iteration; then the same on a grid so small that the kernels
take no time (the host-side floor).

    python tools/rank_rehearsal.py [iters]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from eddy_currents_3d_amd.dist import rccl_rank

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def run(N, sdz, G, r):
    n = N * N * sdz
    with rccl_rank(0, 1, 0, rehearse=(r, G)) as m:
        m.assemble_poisson(N, N, sdz)
        view, k0, k1 = m.slab(0)
        m.upload("B", np.random.Generator(np.random.PCG64(1)).standard_normal(n))
        m.upload("X", np.zeros(n))
        m.iterate_begin()
        m.iterate(1, 20)
        m.synchronize()
        t0 = time.perf_counter()
        m.iterate(21, iters)
        t1 = time.perf_counter()
        m.synchronize()
        t2 = time.perf_counter()
        plan, xd = m.plan()
        calls = m.api_calls(0)
        km = m.iterate(21 + iters, 20, per_kernel=True)
    rows = N * N * (k1 - k0)
    print(f"{N}x{N}x{sdz} on {G} ranks, rank {r} ({k1 - k0} planes, {rows / 2**20:.2f} Mi rows): plan {plan}, X every {xd}: "
          f"{1e3 * (t2 - t0) / iters:.4f} ms per iteration ({rows * iters / (t2 - t0) / 1e9:.2f} G DOF*iters/s on this rank), host "
          f"enqueue {1e3 * (t1 - t0) / iters:.4f} ms per iteration, {calls:.0f} runtime calls per iteration; stages "
          + " ".join(f"{k}={v * 1e3:.0f}us" for k, v in km.items()), flush=True)


def run_av(case, dims, G, r):
    """One rank of an A-V job (BASELINE configs 3 / 5: the shipped geometry resampled to `dims`) alone on this GPU."""
    from eddy_currents_3d_amd import vxc
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", f"g4_{case}.npz"))
    model = vxc.resample(vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                                      tuple(float(x) for x in g["adj"])), *dims)
    t = vxc.domain_tables(model)
    for label, reh in (("undivided", None), (f"rank {r} of {G}", (r, G))):
        ctx = rccl_rank(0, 1, 0, rehearse=reh)
        with ctx as m:
            m.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
            n = m.n
            view, k0, k1 = m.slab(0)
            m.upload("B", np.random.Generator(np.random.PCG64(7)).standard_normal(n))
            m.upload("X", np.zeros(n))
            m.iterate_begin()
            m.iterate(1, 20)
            m.synchronize()
            t0 = time.perf_counter()
            m.iterate(21, iters)
            t1 = time.perf_counter()
            m.synchronize()
            t2 = time.perf_counter()
            plan, xd = m.plan()
            calls = m.api_calls(0)
            km = m.iterate(21 + iters, 20, per_kernel=True)
        print(f"{case} {dims[0]}x{dims[1]}x{dims[2]} (n = {n}), {label} through the RCCL driver ({k1 - k0} planes): plan {plan}, X every "
              f"{xd}: {1e3 * (t2 - t0) / iters:.4f} ms per iteration, host enqueue {1e3 * (t1 - t0) / iters:.4f} ms, {calls:.0f} runtime "
              f"calls per iteration; stages " + " ".join(f"{k}={v * 1e3:.0f}us" for k, v in km.items()), flush=True)


if os.environ.get("REHEARSE_AV"):
    run_av("LIM", (384, 192, 128), 8, 3)
    run_av("LIM", (384, 192, 128), 4, 1)
    if os.environ["REHEARSE_AV"] != "lim":
        run_av("LIM", (384, 192, 128), 2, 1)
        run_av("ec_src_move_hole", (256, 256, 60), 2, 1)
        run_av("ec_src_move_hole", (256, 256, 60), 4, 1)
    sys.exit(0)
if os.environ.get("REHEARSE_ONLY"):
    for tok in os.environ["REHEARSE_ONLY"].split(";"):
        N, sdz, G, r = (int(t) for t in tok.split(","))
        run(N, sdz, G, r)
    sys.exit(0)
for G in (2, 4, 8):
    run(512, 512, G, G // 2)
run(256, 256, 8, 3)
run(64, 128, 8, 3)       # the host-side floor: kernels of a few microseconds
