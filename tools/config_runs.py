#!/usr/bin/env python3
"""BASELINE configs 3 and 5 as whole runs on one MI355X: the shipped geometries resampled to 256x256x60 and
384x192x128 (vxc.resample), device assembly, then 50 / 200 time steps of the reference's loop (source update on the
host, right-hand side, solve, post-update and the field vectors of every output step on the device; no files
written).  Prints wall time, iterations and DOF*iters/s over everything.   python tools/config_runs.py [--slabs N]"""
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
a = ap.parse_args()
for case, dims, steps in (("ec_src_move_hole", (256, 256, 60), 50), ("LIM", (384, 192, 128), 200)):
    g = np.load(os.path.join(REPO, "tests", "golden", f"g4_{case}.npz"))
    small = vxc.VxcModel(g["vox"], [str(s) for s in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    model = vxc.resample(small, *dims)
    solver = E.EC3DSolver() if a.slabs == 1 else E.EC3DMulti(a.slabs, devices=[0] * a.slabs)
    t_solve = [0.0]
    marks = {}

    def on_rhs(k, s, info):
        marks["t"] = time.perf_counter()

    def on_solved(k, s, info):
        t_solve[0] += time.perf_counter() - marks["t"]

    t0 = time.perf_counter()
    with solver as s:
        log = host.run(model, s, steps=steps, out_dir="/tmp/ec3d_cfg", on_rhs=on_rhs, on_solved=on_solved,
                       write_output=lambda N: False)
        n = s.n
    wall = time.perf_counter() - t0
    its = sum(i["iter"] for i in log)
    print(f"{case} {dims[0]}x{dims[1]}x{dims[2]}: n = {n}, {len(log)} time steps, {its} iterations, wall {wall:.2f} s "
          f"(ingest tables + device assembly + loop + field vectors of {len(log) - 1} output steps), in the solver calls "
          f"{t_solve[0]:.2f} s = {n * its / t_solve[0]:.3e} DOF*iters/s; over everything {n * its / wall:.3e}", flush=True)
