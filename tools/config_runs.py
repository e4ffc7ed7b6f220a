#!/usr/bin/env python3
"""BASELINE configs 3 and 5 as whole runs on one MI355X: the shipped geometries resampled to 256x256x60 and
384x192x128 (vxc.resample), device assembly, then 50 / 200 time steps of the reference's loop (source update on the
host, right-hand side, solve, post-update, and field_N.vtk of EVERY output step as the reference writes them,
src/EC3D.f90:436-444).  Prints wall time, iterations, the time inside the solver calls and DOF*iters/s.

    python tools/config_runs.py [--slabs N] [--mode overlap|sync|nofiles] [--out DIR] [--keep]

mode overlap (default): field output beside the next step's solve (host._OutputPipeline); sync: the fields fetched
and the files written inside the loop, as rounds 1-3 did; nofiles: fields brought to the host, nothing written.
Every file is deleted as soon as it is complete unless --keep (200 steps of config 5 are 113 GB)."""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import eddy_currents_3d_amd as E
from eddy_currents_3d_amd import host, vxc

ap = argparse.ArgumentParser()
ap.add_argument("--slabs", type=int, default=1, help="N > 1: the multi-GPU handle with N slabs on device 0")
ap.add_argument("--mode", choices=["overlap", "sync", "nofiles"], default="overlap")
ap.add_argument("--out", default="/tmp/ec3d_cfg")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--cases", default="ec_src_move_hole,LIM")
a = ap.parse_args()
ALL = {"ec_src_move_hole": ((256, 256, 60), 50), "LIM": ((384, 192, 128), 200)}
for case in a.cases.split(","):
    dims, steps = ALL[case]
    g = np.load(os.path.join(REPO, "tests", "golden", f"g4_{case}.npz"))
    small = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    model = vxc.resample(small, *dims)
    solver = E.EC3DSolver() if a.slabs == 1 else E.EC3DMulti(a.slabs, devices=[0] * a.slabs)
    t_solve = [0.0]
    marks = {}
    written = [0, 0]

    parts = {"sources + rhs_step": 0.0, "post_update + output start": 0.0}

    def on_rhs(k, s, info):
        marks["t"] = time.perf_counter()
        if "step_end" in marks:
            parts["sources + rhs_step"] += marks["t"] - marks["step_end"]

    def on_solved(k, s, info):
        marks["solved"] = time.perf_counter()
        t_solve[0] += marks["solved"] - marks["t"]

    def on_step(k, s, info):
        marks["step_end"] = time.perf_counter()
        parts["post_update + output start"] += marks["step_end"] - marks["solved"]

    def on_written(N, paths):
        for p in paths:
            written[0] += 1
            written[1] += os.path.getsize(p)
            if not a.keep:
                os.remove(p)

    t0 = time.perf_counter()
    with solver as s:
        log = host.run(model, s, steps=steps, out_dir=a.out, on_rhs=on_rhs, on_solved=on_solved, on_step=on_step,
                       write_output=(lambda N: False) if a.mode == "nofiles" else None,
                       overlap_output=a.mode != "sync", on_written=on_written)
        n = s.n
    wall = time.perf_counter() - t0
    its = sum(i["iter"] for i in log)
    print(f"{case} {dims[0]}x{dims[1]}x{dims[2]} [{a.mode}{'' if a.slabs == 1 else f', {a.slabs} slabs'}]: n = {n}, {len(log)} time steps, "
          f"{its} iterations, wall {wall:.2f} s (ingest tables + device assembly + loop + output of {len(log) - 1} steps: "
          f"{written[0]} files, {written[1] / 1e9:.1f} GB written), in the solver calls {t_solve[0]:.2f} s = "
          f"{n * its / t_solve[0]:.3e} DOF*iters/s; wall / solver = {wall / t_solve[0]:.2f}; over everything "
          f"{n * its / wall:.3e}; in the loop outside the solver: " +
          ", ".join(f"{k} {v:.2f} s" for k, v in parts.items()), flush=True)
