#!/usr/bin/env python3
"""bench.py — DOF·iters/s of the BiCGSTAB-with-restart hot path on MI355X (fp64).

Contract (driver):  python bench.py --gpus N --steps K --warmup W     -> one JSON line on rank 0.
  * a "step" is one BiCGSTAB iteration = one pass of src/solvers.f90:24-50 (2 SpMV, 5 dots, 2 norms,
    4 vector updates) over the whole grid, convergence exits disabled so exactly K run;
  * workload: synthetic N^3 7-point operator of BASELINE.json configs 2/4 (src/EC3D.f90:528-654 rule,
    delta = 0.00333, BND = -0.95), default 512^3 = the grid the metric's roofline and strong-scaling
    targets are quoted on; it fits one GPU (16 GB), so every N runs the SAME grid: scaling "strong".
    RHS = the deterministic bar source of SURVEY §8c, x0 = 0; operator, b and x are resident in HBM
    before the timed region starts (assembly is on the device, nothing crosses PCIe in the loop);
  * N > 1, two transports behind the same z-slab decomposition (rank g owns planes [g*N/G, (g+1)*N/G)) and ONE invocation
    that runs both:
      - RCCL, one process per GPU (the headline): every rank drives its slab's iteration loop from C++
        (ec3d_multi_create_rank, csrc/ec3d_multi.hip) -- halo planes as ncclSend / ncclRecv pairs on a side stream, the
        eight sums of every rank by ncclAllGather, torch.distributed only hands round the two RCCL ids and the barriers;
      - in-library (sub-record `in_library`): ONE process, N devices behind one handle (include/ec3d_hip.h section 2c),
        one host thread per slab, halo planes pulled over peer access, partial sums read in place.
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (the driver's form): the ranks ARE the RCCL job;
    when its timed region is over and the process group is gone, rank 0 starts the in-library form as a fresh child process.
    Plain `python bench.py --gpus N`: a parent that never initialises a GPU (it only counts devices) starts N fresh rank
    processes, then the in-library child, and prints the one line.  No process that has touched a GPU is ever re-executed.
    Before anything is timed each form checks itself: A*x over the N slabs against the undivided operator on one device bit
    for bit, and ||b|| through the reduction path (`verified`).  A form that fails its check (or dies, or exceeds its time
    limit) is reported as {"error": ...} and the other carries the line; only when both fail is the exit status non-zero.
    The in-library form is retried once with EC3D_MULTI_SERIALIZE_WAITS=1 EC3D_MULTI_FLAT_HUB=1 before it is given up.
    `--gpus 2 --devices 0,0` (rehearsal on one card): the in-library record, and `rccl: {"skipped": ...}` -- RCCL refuses two
    ranks of one communicator on one device, which is known before anything is started.
    Keys of the line beyond the driver's contract when N > 1:
      transport        "rccl" | "in_library": which form the top-level numbers are
      rccl             {nranks (ncclCommCount, as RCCL itself counts the communicator), version (ncclGetVersion), library (file
                       the entry points came from), ranks: [{rank, device, planes, plan, x_update_every, ms_per_step (this
                       rank's own clock around its K steps), stage_us {k1..k5}, reduction_points {us_per_iteration, per_iteration},
                       halo_waits {...}, host {enqueue_ms_per_iteration, api_calls_per_iteration}}, ...]}  | {"skipped"|"error": text}
      in_library       the in-library form's whole line (same keys; its own `verified`), or {"error": text[, "retried_with": ...]}
      verified         text of the check the headline form passed
      host             rank 0's enqueue cost (as before)
  * roofline: the kernel with the largest share of the iteration (measured, not assumed), its
    algorithmic bytes per row from SURVEY §8d / DESIGN.md §4, duration measured live with hipEvents on
    the library's stream inside an extra instrumented pass of the same K iterations; peak 8.0 TB/s
    (MI355X_MICROARCH.md).  With the default dictionary band format K1/K3 move 25/17 B per row instead
    of the 80/72 B of the plain-DIA model, so the widest kernel is K4 (X and R updates, 56 B/row);
    `kernels` lists every stage with both byte models, `spmv` the bare 7-point SpMV the north-star
    target is quoted on.  --format dia runs the plain DIA streams (the SURVEY's byte model).
  * cpu_baseline (rank 0, N = 1 only): the unmodified reference solver (oracle/_ref/ref_solve,
    src/solvers.f90 compiled with amdflang; "port" = our C restatement when that binary is absent)
    on one host core: 20 fixed iterations on the 256^3 cube of BASELINE config 2, its
    wall time reported next to it (`wall_s`); `cpu_baseline_all_cores`: the same sample through the CPU restatement
    under OpenMP on every host core the process may use (thread count in `cores`) -- context, as BASELINE.md section 3 allows.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ITER_BYTES_PER_DOF = 264       # K1 80 + K2 24 + K3 72 + K4 56 + K5 32
# bytes per row each stage has to move: SURVEY §8d model (plain DIA) and the dictionary format
SURVEY_BYTES = {"k1": 80, "k2": 24, "k3": 72, "k4": 56, "k5": 32, "spmv": 72}
DICT_BYTES = {"k1": 25, "k2": 24, "k3": 17, "k4": 56, "k5": 32, "spmv": 17}
KERNEL_NAMES = {"k1": "k1_spmv_dot (AP = A*P, AP.R0)", "k2": "k2_s_update (S = R - alpha*AP, S.S)",
                "k3": "k3_spmv_dots (AS = A*S, AS.S, AS.AS)",
                "k4": "k4_x_r_update (X += alpha*P + omega*S, R = S - omega*AS, R.R, R.R0)",
                "k5": "k5_p_update (P = R + beta*(P - omega*AP))"}


def bar_rhs(N, k0=0, k1=None):
    """mu0*1e6 on the bar i,k in [N/2-2, N/2+3], j in [N/4, 3N/4] (1-based);
    planes k0..k1-1 (0-based) only."""
    import numpy as np
    k1 = N if k1 is None else k1
    mu0 = 0.12566370964050292e-05
    b = np.zeros((k1 - k0, N, N))
    lo, hi = N // 2 - 2, N // 2 + 3
    ka, kb = max(lo - 1, k0), min(hi, k1)
    if kb > ka:
        b[ka - k0:kb - k0, N // 4 - 1:3 * N // 4, lo - 1:hi] = mu0 * 1e6
    return b.reshape(-1)


def av_system(refine):
    """The shipped compare_to_Elmer geometry (tests/golden/g4: plate, hole, two coil pairs), refined by an
    integer factor per axis (physical size kept), as the arrays ec3d_assemble takes, plus a coil RHS:
    the full A-V system [Ax | Ay | Az | U] of BASELINE config 3 at a size worth timing."""
    import numpy as np
    mu0 = 0.12566370964050292e-05
    g = np.load(os.path.join(REPO, "tests", "golden", "g4_compare_to_Elmer.npz"))
    f = int(refine)
    vox = np.repeat(np.repeat(np.repeat(g["vox"], f, axis=0), f, axis=1), f, axis=2)
    dx = float(g["lattice_dim"])
    flat = vox.reshape(-1)
    ncell = flat.size
    geo = flat.astype(np.int8).copy()
    geo[geo == 0] = 6
    geoC = np.zeros(ncell, np.int32)
    idx = np.flatnonzero(flat == 1)
    geoC[idx] = 3 * ncell + 1 + np.arange(idx.size)
    valPHYS = np.zeros((6, 5))
    valPHYS[:, 0] = 1.0
    valPHYS[0, 1] = mu0 * 35.26e6
    b = np.zeros(3 * ncell + idx.size)
    a = 183.0 / (6 * dx * 6 * dx)
    b[np.flatnonzero(flat == 2)] = a * mu0
    b[np.flatnonzero(flat == 3)] = -a * mu0
    b[ncell + np.flatnonzero(flat == 4)] = a * mu0
    b[ncell + np.flatnonzero(flat == 5)] = -a * mu0
    return (geo.reshape(vox.shape), geoC.reshape(vox.shape), valPHYS, np.full((3, 2), -0.95),
            np.array([dx / f] * 3), 1e-3, b)


def av256_system(dims=(256, 256, 256), stem="ec_src_move_hole"):
    """BASELINE config 3 at the size BASELINE.json writes: the shipped ec_src_move_hole geometry (tests/golden/g4:
    plate with a hole under a moving coil pair) resampled to 256 x 256 x 256 with the physical size kept
    (vxc.resample -- the input oracle/make_goldens.py case_g7x ran through the unmodified reference), as the tables
    ec3d_assemble takes plus the first time step's sources (src/EC3D.f90:345-365)."""
    import numpy as np
    from eddy_currents_3d_amd import host, vxc
    g = np.load(os.path.join(REPO, "tests", "golden", f"g4_{stem}.npz"))
    small = vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                         tuple(float(x) for x in g["adj"]))
    model = vxc.resample(small, *dims)
    t = vxc.domain_tables(model)
    idx, val, moving = host.SourceProgram(model, t).step(0.0)
    return model, t, idx, val, moving


def cpu_baseline(N=256, iters=20):
    """Reference solver on one host core, bounded sample (never the thing measured as product): a fixed
    number of iterations on the 256^3 cube of BASELINE config 2, as SURVEY section 8d prescribes for the
    256^3 / 512^3 grids (convergence would take hours)."""
    import numpy as np
    from oracle import oracle as O
    t_wall = time.perf_counter()
    kind = "reference" if O.have_ref() else "port"
    valA, irow, jcol = O.poisson_csr(N, N, N)
    b = bar_rhs(N)
    n = N ** 3
    # itmax = iters - 1: the reference then runs exactly `iters` iterations (src/solvers.f90:25-29).
    # capture_stdout: it prints ||R|| on that exit (:25-28); this script's stdout carries exactly one JSON line
    x, it, sec, _ = O.solve_process(kind, valA, irow, jcol, b, np.zeros(n), 1e-300, iters - 1, capture_stdout=True)
    return {"value": n * it / sec, "unit": "DOF*iters/s", "cores": 1, "kind": kind,
            "sample": f"{N}^3 cube of the same operator/RHS (n={n}, BASELINE config 2 grid), {it} fixed iterations of "
                      f"{'src/solvers.f90 (amdflang -O2)' if kind == 'reference' else 'oracle/ec3d_oracle.c'}"
                      f" in {sec:.2f} s inside the solver call, 1 thread, host cores available: {os.cpu_count()}",
            "solver_s": sec, "wall_s": time.perf_counter() - t_wall}


def cpu_baseline_all_cores(N=256, iters=20):
    """The same sample on ALL host cores this process may use: the CPU restatement under OpenMP (oracle/ec3d_oracle_omp.c --
    the reference itself is serial, src/solvers.f90 has no parallel construct), thread count stated.  Context only
    (BASELINE.md section 3); never the thing measured as product, never a parity checker."""
    import numpy as np
    from oracle import oracle as O
    t_wall = time.perf_counter()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 32))     # (the GPU box gives one GPU's share of the host: 16 cores)
    valA, irow, jcol = O.poisson_csr(N, N, N)
    b = bar_rhs(N)
    n = N ** 3
    old = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OMP_PROC_BIND", "OMP_WAIT_POLICY")}
    os.environ.update(OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="close", OMP_WAIT_POLICY="passive")
    try:
        x, it, sec, _ = O.solve_process("port_omp", valA, irow, jcol, b, np.zeros(n), 1e-300, iters - 1, capture_stdout=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return {"value": n * it / sec, "unit": "DOF*iters/s", "cores": threads, "kind": "port (OpenMP)",
            "sample": f"{N}^3 cube of the same operator/RHS (n={n}), {it} fixed iterations of oracle/ec3d_oracle_omp.c (the CPU "
                      f"restatement of src/solvers.f90 with its loops and dot products under OpenMP; sums in another order than "
                      f"the reference's) in {sec:.2f} s inside the solver call, {threads} threads of {avail} cores available",
            "solver_s": sec, "wall_s": time.perf_counter() - t_wall}


def latest_traffic(grid, fmt, workload, n_gpus):
    """Per-kernel HBM bytes per launch from the committed PMC profile of the SAME configuration
    (profiles/*pmc*.json, written by tools/parse_rocprof.py), or None."""
    pdir = os.path.join(REPO, "profiles")
    try:
        cands = sorted(f for f in os.listdir(pdir) if f.endswith(".json") and "pmc" in f)
    except OSError:
        return None
    for fn in reversed(cands):
        try:
            with open(os.path.join(pdir, fn)) as f:
                tr = json.load(f)
        except (OSError, ValueError):
            continue
        if tr.get("grid") == grid and tr.get("format") == fmt and tr.get("workload", "cube") == workload and \
                tr.get("n_gpus", 1) == n_gpus:
            tr["_file"] = fn
            return tr
    return None


def stage_table(s, info, kernel_ms):
    """Per-stage records of one handle's iteration: which launches it is made of (asked of the library,
    ec3d_get_fusion), each stage's algorithmic bytes per row under this format's and the SURVEY's byte model, its
    measured average and the rate that gives.  Returns (kernels, names, fmt_bytes, dominant stage)."""
    rows = int(info.n)
    fmt_bytes = dict(DICT_BYTES if info.dict_classes > 0 else SURVEY_BYTES)
    survey_bytes = dict(SURVEY_BYTES)
    names = dict(KERNEL_NAMES)
    kernel_ms = dict(kernel_ms)
    k2_in_k3, k5_in_k1 = s.fusion() if hasattr(s, "fusion") else (0, 0)
    if k2_in_k3:
        # K2 runs inside K3 (k23_s_spmv_dots: 2-D tiles, vectors beyond the caches): stage 2 is empty, stage 3
        # forms S = R - alpha*AP where the stencil reads it -- S is written once and not read back: 8 B/row less
        kernel_ms["k3"] += kernel_ms.pop("k2")
        fmt_bytes["k3"] += fmt_bytes.pop("k2") - 8
        survey_bytes["k3"] += survey_bytes.pop("k2") - 8
        names["k3"] = "k23_s_spmv_dots (S = R - alpha*AP inside AS = A*S; S.S, AS.S, AS.AS)"
    if k5_in_k1:
        # K5 runs inside the NEXT iteration's K1 (k51_p_spmv_dot): stage 1 is empty after the first iteration,
        # stage 5 forms P where the stencil of AP = A*P reads it -- P is written once and not read back
        kernel_ms["k5"] += kernel_ms.pop("k1")
        fmt_bytes["k5"] += fmt_bytes.pop("k1") - 8
        survey_bytes["k5"] += survey_bytes.pop("k1") - 8
        names["k5"] = "k51_p_spmv_dot (P = R + beta*(P - omega*AP) inside the next AP = A*P; AP.R0)"
    # What K4 moves when X is updated every D-th iteration (ec3d_get_x_interval) and when it runs as an SpMV kernel that
    # computes A*S again (ec3d_get_k4_form): D - 1 launches without X and one that applies D updates, averaged.
    # bytes_per_row is what the launches are built to move; survey_bytes_per_row stays SURVEY 8d's 56 B.
    D = s.x_interval() if hasattr(s, "x_interval") else 1
    k4s = bool(s.k4_as_spmv()) if hasattr(s, "k4_as_spmv") else False
    xg = s.x_groups()[0] if hasattr(s, "x_groups") else 0
    if k4s:
        fmt_bytes["k3"] -= 8                       # K2-in-K3 no longer writes AS
        off, on = 25, 33 + 16 * D                  # S + class byte + R0 read, R written; + X, D P, D - 1 older S; X written
        names["k4"] = (f"k4s_x_r_spmv (K4 as an SpMV kernel: AS = A*S computed again; X = X + alpha*P + omega*S applied "
                       f"every {D} iterations, in order)")
        if int(xg) == 2:
            # every K4 the light launch; each group of D updates by a streaming launch of its own on the same stream
            # (k_x_group: X, D P, D S read, X written), timed with the K4 of the group's last iteration
            on = 25 + 16 + 16 * D
            names["k4"] = (f"k4s_x_r_spmv (K4 as an SpMV kernel: AS = A*S computed again) + k_x_group every {D} iterations "
                           f"(X = X + alpha*P + omega*S of the group, in order, a launch of its own on the same stream)")
    else:
        off, on = 32, 40 + 16 * D
        if D > 1:
            names["k4"] = f"k4d_x_r_update (K4 with X = X + alpha*P + omega*S applied every {D} iterations, in order)"
    if k4s or D > 1:
        fmt_bytes["k4"] = ((D - 1) * off + on) / D
    total = sum(kernel_ms.values())
    kernels = {}
    for k, ms in kernel_ms.items():
        kernels[k] = {"name": names[k], "ms": ms, "share": ms / total,
                      "bytes_per_row": fmt_bytes[k], "GBps": fmt_bytes[k] * rows / ms / 1e6,
                      "survey_bytes_per_row": survey_bytes[k],
                      "survey_GBps": survey_bytes[k] * rows / ms / 1e6}
    dom = max(kernel_ms, key=kernel_ms.get)   # dominant kernel by measured share
    return kernels, names, fmt_bytes, dom


def side_workload(E, name, device, K=200, W=5):
    """A driver-timed figure for a workload other than the headline one, on a fresh handle, AFTER the headline's
    timed region: the same measurement (W warm-up iterations, K timed between two synchronisations, then an
    instrumented pass with hipEvents at every kernel boundary), condensed to one sub-record.
      av       the full A-V system [Ax | Ay | Az | U] (the matrix src/EC3D.f90:408 solves) of the shipped
               compare_to_Elmer geometry refined x3 per axis: 306 x 306 x 72, n = 21.4 M
      av256    BASELINE config 3 at its stated size: ec_src_move_hole resampled to 256 x 256 x 256, n = 50.3 M + U --
               400 MB per vector, beyond the 256 MiB Infinity Cache: the reference's own matrix on the HBM roofline
      cube256  the 256^3 cube of BASELINE config 2"""
    import numpy as np
    import torch
    t_wall = time.perf_counter()
    with E.EC3DSolver(device=device) as s:
        b = None
        if name == "av":
            geo, geoC, valPHYS, BND, delta, dt, b = av_system(3)
            s.assemble(geo, geoC, valPHYS, BND, delta, dt)
            n = len(b)
            what = (f"full A-V system of the shipped compare_to_Elmer geometry refined x3: grid "
                    f"{geo.shape[2]}x{geo.shape[1]}x{geo.shape[0]}, {int(np.count_nonzero(geoC))} conducting cells, "
                    f"coil RHS, x0=0, exits disabled")
        elif name == "av256":
            model, t, idx, val, moving = av256_system()
            s.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
            n = s.n
            sdz, sdy, sdx = model.vox.shape
            what = (f"BASELINE config 3 at its stated size: full A-V system [Ax|Ay|Az|U] of the shipped ec_src_move_hole "
                    f"geometry resampled to {sdx}x{sdy}x{sdz} (physical size kept; the input of tests/golden/g7x_*), "
                    f"{int(t['ncells0'])} conducting cells, first time step's coil RHS (src/EC3D.f90:345-404), x0=0, "
                    f"exits disabled")
        else:
            N = 256
            s.assemble_poisson(N, N, N)
            n = N ** 3
            b = bar_rhs(N)
            what = "synthetic 256^3 7-pt operator (BASELINE config 2 grid), bar RHS, x0=0, exits disabled"
        s.upload("X", np.zeros(n))
        if name == "av256":
            s.upload("B", np.zeros(n))
            s.rhs_step(idx, val, moving=moving)     # the reference's Jaf of the first step, built on the device
        else:
            s.upload("B", b)
        s.iterate_begin()
        s.iterate(1, W)
        s.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate(W + 1, K)
        s.synchronize()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # two instrumented passes, the smaller average of each stage: at these sizes a stage's time depends on what the
        # previous one left in the 256 MiB Infinity Cache, and a pass of hipEvents at every kernel boundary disturbs that
        # more on some runs than on others (K4 at 21 M unknowns: 165 ... 190 us; rocprofv3, which needs no events in the
        # stream, sees 165: profiles/r04_av_kernel_stats.csv)
        km1 = s.iterate(W + K + 1, 50, per_kernel=True)
        km2 = s.iterate(W + K + 51, 50, per_kernel=True)
        kernel_ms = {k: min(km1[k], km2[k]) for k in km1}
        spmv_ms = s.time_kernel("spmv", 20)
        info = s.info
        kernels, names, fmt_bytes, dom = stage_table(s, info, kernel_ms)
        geom = {"vector": int(s.geometry(0).nblk), "spmv": int(s.geometry(1).nblk)}
        vplace = s.vector_placement() if hasattr(s, "vector_placement") else ([], -1, 0.0)
    rows = int(info.n)
    achieved = fmt_bytes[dom] * rows / (kernels[dom]["ms"] * 1e-3) / 1e9
    per_iter = sum(fmt_bytes[k] for k in kernels)
    return {"workload": what, "n": n, "value": n * K / elapsed, "unit": "DOF*iters/s", "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "workgroups": geom,
            "bytes_per_dof_iter": per_iter, "iter_hbm_frac": per_iter * n * K / elapsed / 1e9 / PEAK_HBM_GBS,
            "kernels": {k: {"ms": v["ms"], "bytes_per_row": v["bytes_per_row"], "GBps": v["GBps"],
                            "frac": v["GBps"] / PEAK_HBM_GBS} for k, v in kernels.items()},
            "dominant": {"kernel": names[dom], "bytes_per_row": fmt_bytes[dom], "avg_launch_ms": kernels[dom]["ms"],
                         "achieved": achieved, "frac": achieved / PEAK_HBM_GBS},
            "spmv": {"ms": spmv_ms, "bytes_per_row": fmt_bytes["spmv"], "GBps": fmt_bytes["spmv"] * rows / spmv_ms / 1e6,
                     "frac": fmt_bytes["spmv"] * rows / spmv_ms / 1e6 / PEAK_HBM_GBS},
            **({"vector_placement": {"candidate_us_per_iteration": [round(v, 1) for v in vplace[0]], "kept": vplace[1],
                                     "search_ms": round(vplace[2], 1)}} if vplace[0] else {}),
            "wall_s": time.perf_counter() - t_wall}


# ---------------------------------------------------------------------------------------------------------------
# N > 1: child processes.  Nothing here touches a GPU; every child is a fresh `python bench.py ...` process.
PASS_FLAGS = ("steps", "warmup", "grid", "format", "workload", "refine", "cpu_grid", "cpu_iters")


def child_argv(args, gpus, extra):
    argv = [sys.executable, os.path.abspath(__file__), "--gpus", str(gpus)]
    for k in PASS_FLAGS:
        argv += ["--" + k.replace("_", "-"), str(getattr(args, k))]
    if args.no_verify:
        argv.append("--no-verify")
    return argv + ["--no-cpu-baseline", "--no-side-workloads", "--no-spmv-dia"] + list(extra)


def last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None


def run_in_library_child(args, devices, timeout_s):
    """Today's one-process form as a fresh child: its whole JSON line, or {"error": ...}; retried once with the
    conservative synchronisation (EC3D_MULTI_SERIALIZE_WAITS=1 EC3D_MULTI_FLAT_HUB=1) before it is given up."""
    import subprocess
    argv = child_argv(args, len(devices), ["--role", "inlib", "--devices", ",".join(str(d) for d in devices)])
    first_error = None
    for attempt, extra_env in enumerate(({}, {"EC3D_MULTI_SERIALIZE_WAITS": "1", "EC3D_MULTI_FLAT_HUB": "1"})):
        env = dict(os.environ, **extra_env)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID"):
            env.pop(k, None)     # (under the launcher: the child is not a rank of that job)
        try:
            p = subprocess.run(argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            rec = last_json_line(p.stdout.decode(errors="replace"))
            if p.returncode == 0 and rec is not None:
                if attempt:
                    rec["retried_with"] = extra_env
                    rec["first_attempt_error"] = first_error
                return rec
            err = (f"exit status {p.returncode}: " + p.stderr.decode(errors="replace").strip()[-600:])
        except subprocess.TimeoutExpired:
            err = f"no result within {timeout_s} s (killed)"
        if first_error is None:
            first_error = err
    return {"error": first_error, "retried_with": {"EC3D_MULTI_SERIALIZE_WAITS": "1", "EC3D_MULTI_FLAT_HUB": "1"},
            "retry_error": err}


def run_rccl_ranks(args, devices, timeout_s):
    """N fresh rank processes of the RCCL driver (what the launcher would start), rank r on device devices[r]; rank 0's
    JSON line, or {"error": ...}.  A rank that ends with an error ends the others (their exact PIDs)."""
    import socket
    import subprocess
    import tempfile
    N = len(devices)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    argv = child_argv(args, N, ["--role", "rank", "--no-in-library"])
    logdir = tempfile.mkdtemp(prefix="ec3d_bench_ranks_")
    procs = []
    for r, d in enumerate(devices):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(N), LOCAL_RANK=str(d), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        err = open(os.path.join(logdir, f"rank{r}.err"), "wb")
        procs.append((subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=err), err))
    t0 = time.time()
    failed = None
    while True:
        codes = [p.poll() for p, _ in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0]} ended with exit status {codes[bad[0]]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > timeout_s:
            failed = f"no result within {timeout_s} s"
            break
        time.sleep(0.2)
    out0 = b""
    if failed:
        for p, _ in procs:
            if p.poll() is None:
                p.kill()
    try:
        out0 = procs[0][0].communicate(timeout=30)[0] or b""
    except Exception:      # noqa: BLE001 -- reporting only
        pass
    for p, err in procs:
        try:
            p.wait(timeout=30)
        except Exception:  # noqa: BLE001
            pass
        err.close()
    rec = last_json_line(out0.decode(errors="replace")) if not failed else None
    if rec is None:
        tails = []
        for r in range(N):
            try:
                with open(os.path.join(logdir, f"rank{r}.err"), "rb") as f:
                    t = f.read().decode(errors="replace").strip()
                if t:
                    tails.append(f"[rank {r}] " + t[-400:])
            except OSError:
                pass
        return {"error": (failed or "rank 0 printed no line") + ("; " + " | ".join(tails) if tails else "")}
    return rec


def merge_forms(rccl_rec, inlib_rec):
    """The one line of an N > 1 invocation from its two forms: RCCL carries it when it produced numbers, else the in-library
    form; the other travels as a sub-record.  Returns (line, exit status)."""
    rccl_ok = isinstance(rccl_rec, dict) and "value" in rccl_rec
    inlib_ok = isinstance(inlib_rec, dict) and "value" in inlib_rec
    if rccl_ok:
        out = rccl_rec
        out["transport"] = "rccl"
        if inlib_rec is not None:
            out["in_library"] = inlib_rec
        return out, 0
    if inlib_ok:
        out = dict(inlib_rec)
        out["transport"] = "in_library"
        out["rccl"] = rccl_rec if rccl_rec is not None else {"skipped": "not attempted"}
        out["in_library"] = {k: inlib_rec[k] for k in ("value", "ms_per_step", "verified", "retried_with", "first_attempt_error")
                             if k in inlib_rec}
        return out, 0
    return {"metric": "DOF*iters/s (fp64 BiCGSTAB-with-restart, 7-pt A-V operator)", "value": None, "unit": "DOF*iters/s",
            "rccl": rccl_rec, "in_library": inlib_rec, "error": "both multi-GPU forms failed"}, 1


def parent(args):
    """Plain `python bench.py --gpus N [--devices ...]`: start the two forms as fresh processes; never touch a GPU here."""
    import torch
    N = args.gpus
    if args.devices is not None:
        devices = [int(t) for t in args.devices.split(",")]
        if len(devices) != N:
            raise SystemExit(f"bench.py: --devices names {len(devices)} devices for --gpus {N}")
    else:
        devices = list(range(N))
    have = torch.cuda.device_count()     # counting does not initialise the GPU
    if max(devices) >= have:
        raise SystemExit(f"bench.py --gpus {N} needs {max(devices) + 1} devices, this machine has {have}")
    if len(set(devices)) < N:
        rccl_rec = {"skipped": f"RCCL (like NCCL) refuses two ranks of one communicator on one device (devices {devices}): "
                               f"known before anything is started, so no rank process was -- the in-library form below runs "
                               f"the same slabs, plans and kernels with local copies in place of the transport"}
    elif args.workload != "cube":
        rccl_rec = {"skipped": "the one-process-per-GPU form of this script runs the cube workloads"}
    else:
        rccl_rec = run_rccl_ranks(args, devices, args.child_timeout)
    inlib_rec = None if args.no_in_library else run_in_library_child(args, devices, args.child_timeout)
    out, status = merge_forms(rccl_rec, inlib_rec)
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    raise SystemExit(status)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300,
                    help="timed iterations (default 300: the GPU leg then lasts about a second at 512^3)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=512, help="cube edge N (512 = headline, 256 = config 2)")
    ap.add_argument("--format", choices=["dict", "dia"], default="dict",
                    help="band storage: dictionary (default, 1 B/row) or plain DIA streams (56 B/row)")
    ap.add_argument("--workload", choices=["cube", "av", "av256"], default="cube",
                    help="cube: the synthetic N^3 operator the metric is quoted on (default); av: the full A-V "
                         "system of the shipped compare_to_Elmer geometry refined by --refine; av256: BASELINE config 3 at "
                         "its stated size, ec_src_move_hole resampled to 256^3 (n = 53.2 M; 1 GPU only)")
    ap.add_argument("--refine", type=int, default=3)
    ap.add_argument("--force-dist", action="store_true",
                    help="use the z-slab/torch.distributed path even with one rank (rehearsal on one GPU)")
    ap.add_argument("--rehearse", type=str, default=None, metavar="R,G",
                    help="with --force-dist on ONE GPU: run rank R of a G-rank job alone (its slab, plan, launches and RCCL "
                         "calls, neighbours mapped to itself): what one rank's iteration costs, host enqueue included")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-spmv-dia", action="store_true", help="skip the plain-DIA SpMV figure after the timed region")
    ap.add_argument("--no-side-workloads", action="store_true",
                    help="skip the A-V (refine 3) and 256^3 sub-records timed after the headline region")
    ap.add_argument("--no-verify", action="store_true",
                    help="in-library multi-GPU path: skip the A*x / reduction check against one device before timing")
    ap.add_argument("--cpu-grid", type=int, default=256, help="cube edge of the cpu_baseline sample")
    ap.add_argument("--cpu-iters", type=int, default=20, help="fixed iteration count of the cpu_baseline sample")
    ap.add_argument("--devices", type=str, default=None,
                    help="in-library multi-GPU path: comma-separated device ordinals, one per slab; a device may "
                         "repeat (rehearsal of N slabs on one card, e.g. --gpus 2 --devices 0,0)")
    ap.add_argument("--role", choices=["rank", "inlib"], default=None,
                    help="(set by this script for its child processes) rank: one rank of the RCCL job, as the launcher starts "
                         "it; inlib: the in-library form in this one process")
    ap.add_argument("--no-in-library", action="store_true", help="N > 1: skip the in-library form's sub-record")
    ap.add_argument("--child-timeout", type=int, default=420, help="N > 1: seconds a child form may take")
    args = ap.parse_args()
    if args.rehearse:
        args.force_dist = True
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if args.role is None and not under_launcher and not args.force_dist and (args.gpus > 1 or args.devices is not None):
        parent(args)      # (does not return)

    # This script's stdout carries exactly one JSON line.  Native libraries write there too (RCCL prints its
    # version banner when the box exports NCCL_DEBUG=VERSION; the reference solver prints ||R|| on the itmax
    # exit), so file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved one.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch  # first: the library then shares torch's HIP runtime
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the in-library form: N devices behind one handle, this one process (a child of parent() / of the launcher's rank 0)
    in_library = args.role == "inlib"
    devices = None
    if in_library:
        if args.devices is not None:
            devices = [int(t) for t in args.devices.split(",")]
            if len(devices) != args.gpus:
                raise SystemExit(f"bench.py: --devices names {len(devices)} devices for --gpus {args.gpus}")
        have = torch.cuda.device_count()     # counting does not initialise the GPU
        if (devices is None and have < args.gpus) or (devices and max(devices) >= have):
            raise SystemExit(f"bench.py --gpus {args.gpus} needs {args.gpus} devices, this machine has {have}")
    elif world != args.gpus:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU path)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist or args.role == "rank"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    import numpy as np
    import eddy_currents_3d_amd as E

    N, K, W = args.grid, args.steps, args.warmup
    n_global = N ** 3
    kernel_ms = None
    workload = (f"synthetic {N}^3 7-pt operator (BASELINE config {'4' if N == 512 else '2'} grid), bar RHS, "
                f"x0=0, exits disabled")
    grid = [N, N, N]
    if args.workload == "av256" and (use_dist or in_library):
        raise SystemExit("bench.py --workload av256 runs on one GPU")
    if args.workload == "av" and use_dist:
        raise SystemExit("bench.py --workload av runs on one GPU or on the in-library multi-GPU path")
    x_group_mode, vplace = 0, ([], -1, 0.0)
    if in_library:
        G = args.gpus
        s = E.EC3DMulti(G, devices=devices, dictionary=args.format == "dict")
        if args.workload == "av":
            geo, geoC, valPHYS, BND, delta, dt, b = av_system(args.refine)
            s.assemble(geo, geoC, valPHYS, BND, delta, dt)
            n_global = len(b)
            grid = list(geo.shape[::-1])
            workload = (f"full A-V system [Ax|Ay|Az|U] of the shipped compare_to_Elmer geometry refined x"
                        f"{args.refine} per axis (BASELINE config 3 style): grid {grid[0]}x{grid[1]}x{grid[2]}, "
                        f"{int(np.count_nonzero(geoC))} conducting cells, coil RHS, x0=0, exits disabled")
            s.upload("B", b)
        else:
            s.assemble_poisson(N, N, N)
            s.upload("B", bar_rhs(N))
        verified = None
        if not args.no_verify:
            # Before anything is timed: the transport between the devices must be RIGHT, not only fast.  A*x over the
            # G slabs (slab operators + halo planes pulled over peer access) against one device, bit for bit, and the
            # reduction path (every rank's sums read in place) through the true residual of (b, x).
            xs = np.sin(0.37 * np.arange(n_global, dtype=np.float64))
            y_multi = s.spmv(xs)
            s.upload("X", xs)
            res_multi = s.true_residual()
            with E.EC3DSolver(device=(devices or [0])[0], dictionary=args.format == "dict") as one:
                if args.workload == "av":
                    one.assemble(geo, geoC, valPHYS, BND, delta, dt)
                    one.upload("B", b)
                else:
                    one.assemble_poisson(N, N, N)
                    one.upload("B", bar_rhs(N))
                y_one = one.spmv(xs)
                one.upload("X", xs)
                res_one = one.true_residual()
            if not np.array_equal(y_multi, y_one):
                bad = int(np.count_nonzero(y_multi != y_one))
                raise SystemExit(f"bench.py: A*x over {G} devices differs from one device in {bad} of {n_global} rows "
                                 f"-- the halo exchange between the GPUs is broken; no number reported")
            if abs(res_multi[0] - res_one[0]) > 1e-9 * abs(res_one[0]) or abs(res_multi[1] - res_one[1]) > 1e-9 * res_one[1]:
                raise SystemExit(f"bench.py: reductions over {G} devices give {res_multi}, one device {res_one}; "
                                 f"no number reported")
            verified = (f"A*x over {G} devices == one device bit for bit ({n_global} rows); ||b - A x||/||b|| and ||b|| "
                        f"equal to {abs(res_multi[0] - res_one[0]) / abs(res_one[0]):.1e} / "
                        f"{abs(res_multi[1] - res_one[1]) / res_one[1]:.1e}")
            del xs, y_multi, y_one
        s.upload("X", np.zeros(n_global))
        s.iterate_begin()
        s.iterate(1, W)
        s.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate(W + 1, K)
        s.synchronize()
        for d in sorted(set(devices or range(G))):
            torch.cuda.synchronize(d)
        t1 = time.perf_counter()
        elapsed = t1 - t0
        kernel_ms = s.iterate(W + K + 1, min(K, 20), per_kernel=True)   # rank 0's stages
        spmv_ms = None
        slab0 = s.slab(0)[0]
        geom = {"vector": int(slab0.geometry(0).nblk), "spmv": int(slab0.geometry(1).nblk)}
        info = slab0.info
        world = G
        multi_plan, multi_xd = s.plan()
        fusion_state = slab0.fusion()
        x_every = multi_xd
        k4_spmv = bool(slab0.k4_as_spmv())
        parallelism = (f"z-slab x{G} inside the library (one process, one host thread per slab, halo planes pulled "
                       f"over peer access, partial sums read in place), devices {devices or list(range(G))}; "
                       + {0: "five launches per iteration, P and S exchanged in front of K1 / K3",
                          1: "five launches per iteration, K1 / K3 as interior + boundary launch around the exchange of P / S",
                          2: "five launches per iteration, K2 / K5 boundary tiles first",
                          5: "five launches per iteration, K2 / K5 boundary planes first and K1 / K3 interior planes first",
                          3: "three launches per iteration (K2 inside K3, K4 as an SpMV kernel, K5 inside the next K1), "
                             "AP and R exchanged",
                          4: "three launches per iteration (K2 inside K3, K4 as an SpMV kernel, K5 inside the next K1), AP and "
                             "R exchanged behind the boundary launches of their producers"}[multi_plan])
    elif not use_dist:
        s = E.EC3DSolver(device=local_rank, dictionary=args.format == "dict")
        if args.workload == "av":
            geo, geoC, valPHYS, BND, delta, dt, b = av_system(args.refine)
            s.assemble(geo, geoC, valPHYS, BND, delta, dt)
            n_global = len(b)
            grid = list(geo.shape[::-1])
            workload = (f"full A-V system [Ax|Ay|Az|U] of the shipped compare_to_Elmer geometry refined x"
                        f"{args.refine} per axis (BASELINE config 3 style): grid {grid[0]}x{grid[1]}x{grid[2]}, "
                        f"{int(np.count_nonzero(geoC))} conducting cells, coil RHS, x0=0, exits disabled")
            s.upload("B", b)
        elif args.workload == "av256":
            model, t, idx, val, moving = av256_system()
            s.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
            n_global = s.n
            grid = list(model.vox.shape[::-1])
            workload = (f"BASELINE config 3 at its stated size: full A-V system [Ax|Ay|Az|U] of the shipped ec_src_move_hole "
                        f"geometry resampled to {grid[0]}x{grid[1]}x{grid[2]} (physical size kept), {int(t['ncells0'])} "
                        f"conducting cells, first time step's coil RHS, x0=0, exits disabled")
            s.upload("B", np.zeros(n_global))
            s.rhs_step(idx, val, moving=moving)
        else:
            s.assemble_poisson(N, N, N)
            s.upload("B", bar_rhs(N))
        s.upload("X", np.zeros(n_global))
        s.iterate_begin()
        s.iterate(1, W)
        s.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate(W + 1, K)
        s.synchronize()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        elapsed = t1 - t0
        # instrumented pass of the same iterations (at most 50 of them: six hipEvents each), library stream
        kernel_ms = s.iterate(W + K + 1, min(K, 50), per_kernel=True)
        spmv_ms = s.time_kernel("spmv", 20)
        geom = {"vector": int(s.geometry(0).nblk), "spmv": int(s.geometry(1).nblk)}
        info = s.info
        fusion_state = s.fusion()
        x_every = s.x_interval() if hasattr(s, "x_interval") else 1
        k4_spmv = bool(s.k4_as_spmv()) if hasattr(s, "k4_as_spmv") else False
        x_group_mode = int(s.x_groups()[0]) if hasattr(s, "x_groups") else 0
        vplace = s.vector_placement() if hasattr(s, "vector_placement") else ([], -1, 0.0)
        parallelism = "single GPU"
    else:
        # one process per GPU: this rank's slab, the whole iteration loop enqueued from C++, RCCL between the ranks
        # (halo planes as ncclSend / ncclRecv on a side stream, the sums by ncclAllGather; csrc/ec3d_multi.hip)
        from eddy_currents_3d_amd.dist import rccl_rank
        rehearse = None
        if args.rehearse:
            rehearse = tuple(int(t) for t in args.rehearse.split(","))
            if world != 1:
                raise SystemExit("bench.py --rehearse R,G runs as ONE process (one rank of a G-rank job on one GPU)")
        s = rccl_rank(rank, world, local_rank, dictionary=args.format == "dict", rehearse=rehearse)
        s.assemble_poisson(N, N, N)
        s.upload("B", bar_rhs(N))
        view, k0, k1 = s.slab(0)
        verified = None
        if not args.no_verify and not rehearse:
            # Before anything is timed: this rank's part of A*x, with the halo planes coming from the z-neighbours
            # over RCCL, against the undivided operator on this rank's own GPU, bit for bit; and ||b|| as every rank
            # derives it from the all-gathered sums against the value computed on the host.
            xs = np.sin(0.37 * np.arange(n_global, dtype=np.float64))
            kd = N * N
            ap = s.spmv(xs)[k0 * kd:k1 * kd]
            s.upload("X", xs)
            res_multi = s.true_residual()
            with E.EC3DSolver(device=local_rank, dictionary=args.format == "dict") as one:
                one.assemble_poisson(N, N, N)
                y_one = one.spmv(xs)[k0 * kd:k1 * kd]
            bad = int(np.count_nonzero(ap != y_one))
            bn, bn_host = res_multi[1], float(np.linalg.norm(bar_rhs(N)))
            flag = torch.tensor([bad, int(abs(bn - bn_host) > 1e-12 * bn_host)], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag)
            if int(flag[0]) or int(flag[1]):
                # every rank sees the same flags: all leave together; rank 0 reports the failure and lets the in-library
                # form carry the line (exit status 0 only if that one verifies)
                text = (f"A*x over {world} ranks differs from one GPU in {int(flag[0])} rows (rank {rank}: {bad} of its slab), "
                        f"||b|| from the all-gathered sums {bn!r} vs {bn_host!r} on {int(flag[1])} ranks -- the exchange over "
                        f"RCCL is broken; nothing timed on this transport")
                s.close()
                dist.barrier()
                dist.destroy_process_group()
                if rank != 0:
                    raise SystemExit(0)
                inlib = None if args.no_in_library else run_in_library_child(args, list(range(world)), args.child_timeout)
                line, status = merge_forms({"error": text}, inlib)
                os.write(real_stdout, (json.dumps(line) + "\n").encode())
                raise SystemExit(status)
            verified = (f"A*x over {world} ranks (halo planes over RCCL) == one GPU bit for bit on every slab; ||b|| from "
                        f"the all-gathered sums equal to {abs(bn - bn_host) / bn_host:.1e}")
            del xs, ap, y_one
        s.upload("X", np.zeros(n_global))
        s.iterate_begin()
        s.iterate(1, W)
        s.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.iterate(W + 1, K)
        t_enq = time.perf_counter() - t0
        s.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        el = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        api_calls = s.api_calls(0)          # of the timed call (the instrumented pass below adds its events)
        # instrumented pass: this rank's stages, and how long its compute stream stood at the reduction points and at the
        # waits for halo planes (events on the compute stream around each: ec3d_multi_iterate_timed)
        kernel_ms, sync = s.iterate_timed(W + K + 1, min(K, 20), 0)
        spmv_ms = None
        geom = {"vector": int(view.geometry(0).nblk), "spmv": int(view.geometry(1).nblk)}
        info = view.info
        multi_plan, multi_xd = s.plan()
        fusion_state = view.fusion()
        x_every = multi_xd
        k4_spmv = bool(view.k4_as_spmv())
        host_enqueue_ms = t_enq * 1e3 / K
        # what RCCL itself says about the job, and every rank's own account (all ranks, not rank 0 only)
        rc_n, rc_v, rc_path = s.rccl_info()
        mine = {"rank": rank, "device": local_rank, "planes": [int(k0), int(k1)], "plan": int(multi_plan),
                "x_update_every": int(multi_xd), "ms_per_step": (t1 - t0) * 1e3 / K,
                "stage_us": {k: v * 1e3 for k, v in kernel_ms.items()},
                "reduction_points": {"us_per_iteration": sync["reduction_points"][0] * 1e3,
                                     "per_iteration": sync["reduction_points"][1]},
                "halo_waits": {"us_per_iteration": sync["halo_waits"][0] * 1e3, "per_iteration": sync["halo_waits"][1]},
                "host": {"enqueue_ms_per_iteration": host_enqueue_ms, "api_calls_per_iteration": api_calls},
                "rccl_nranks": int(rc_n)}
        all_ranks = [None] * world
        dist.all_gather_object(all_ranks, mine)
        rccl_record = {"nranks": int(rc_n), "version": int(rc_v), "library": rc_path,
                       "nranks_agreed": all(r["rccl_nranks"] == rc_n for r in all_ranks), "ranks": all_ranks,
                       "note": "nranks = ncclCommCount of the communicator the sums travel on; stage_us / reduction_points / "
                               "halo_waits from an instrumented pass after the timed region (events on each rank's compute "
                               "stream); ms_per_step is each rank's own clock around the timed iterations"}
        if rehearse:
            # one rank of a G-rank job alone on this GPU: its rows x K iterations (NOT the job's throughput)
            n_global = int(info.n)
            workload += (f"; REHEARSAL of rank {rehearse[0]} of {rehearse[1]} on one GPU: its slab ({k1 - k0} planes), plan, "
                         f"launches and RCCL calls, neighbours mapped to itself -- a timing of one rank's iteration, not a "
                         f"solve")
        parallelism = (f"z-slab x{rehearse[1] if rehearse else world}, one process per GPU, iteration loop in C++, RCCL: halo "
                       f"planes as ncclSend/ncclRecv on a side stream, the sums by ncclAllGather; "
                       + {0: "five launches per iteration", 1: "five launches per iteration, K1 / K3 split around the exchange",
                          2: "five launches per iteration, K2 / K5 boundary tiles first",
                          5: "five launches per iteration, K2 / K5 boundary planes first and K1 / K3 interior planes first",
                          3: "three launches per iteration, AP and R exchanged",
                          4: "three launches per iteration, AP and R exchanged behind the boundary launches of their "
                             "producers"}[multi_plan])


    class _Fusion:       # the headline handle's launches per iteration
        def __init__(self, st, d, k4s, xg=0):
            self.st, self.d, self.k4s, self.xg = st, d, k4s, xg

        def x_groups(self):
            return self.xg, 0

        def fusion(self):
            return self.st

        def x_interval(self):
            return self.d

        def k4_as_spmv(self):
            return self.k4s
    fusion_of = _Fusion(fusion_state, x_every, k4_spmv, x_group_mode)

    # The north-star SpMV figure in the driver-run line: the plain 7-band DIA SpMV (56 B of coefficients + x + y =
    # 72 B/row) at the same grid, timed after the headline region on a handle of its own (the
    # default format's handle is closed first: 7.5 GB of bands + 8.6 GB of vectors), 20 launches back to back.
    spmv_dia = None
    iter_dia = None
    if rank == 0 and not use_dist and not in_library and args.workload == "cube" and args.format == "dict" \
            and not args.no_spmv_dia:
        s.close()
        # (X = X + alpha*P + omega*S in EVERY iteration on this handle: SURVEY section 8d's K4 moves 56 B per row)
        xd_env = os.environ.get("EC3D_XDEFER")
        os.environ["EC3D_XDEFER"] = "1"
        with E.EC3DSolver(device=local_rank, dictionary=False) as sd:
            sd.assemble_poisson(N, N, N)
            sd.upload("B", bar_rhs(N))
            sd.upload("X", np.zeros(n_global))
            ms_d = sd.time_kernel("spmv", 20)
            cand_us, kept = sd.band_placement() if hasattr(sd, "band_placement") else ([], -1)
            spmv_dia = {"kernel": "k_spmv, plain DIA (7 fp64 coefficient streams + x + y)", "ms": ms_d,
                        "placement": {"candidate_us": [round(v, 1) for v in cand_us], "kept": kept},
                        "bytes_per_row": 72, "GBps": 72 * n_global / ms_d / 1e6,
                        "frac": 72 * n_global / ms_d / 1e6 / PEAK_HBM_GBS, "workgroups": int(sd.geometry(1).nblk),
                        "note": "20 launches back to back after the timed region; the time depends on where the driver "
                                "put the 7.5 GB of band streams (1.68 ... 2.0 ms at 512^3 from one allocation to the "
                                "next), so the library looks at up to 8 placements at set-up and keeps the fastest "
                                "(DESIGN.md section 4, profiles/r03_dia_placement.log)"}
            # The WHOLE iteration on SURVEY section 8d's exact byte model, driver-timed: plain DIA streams, five launches
            # (K1 80 + K2 24 + K3 72 + K4 56 + K5 32 = 264 B per DOF*iter), X updated in every iteration.
            Kd, Wd = 200, 5
            sd.iterate_begin()
            sd.iterate(1, Wd)
            sd.synchronize()
            torch.cuda.synchronize()
            td0 = time.perf_counter()
            sd.iterate(Wd + 1, Kd)
            sd.synchronize()
            torch.cuda.synchronize()
            td = time.perf_counter() - td0
            kd_ms = sd.iterate(Wd + Kd + 1, 50, per_kernel=True)
            iter_dia = {"workload": f"the same {N}^3 operator as seven plain fp64 band streams, five launches per iteration, "
                                    f"X updated every iteration: SURVEY section 8d's byte model exactly",
                        "byte_model": "survey_8d", "steps": Kd, "warmup": Wd, "ms_per_step": td * 1e3 / Kd,
                        "value": n_global * Kd / td, "unit": "DOF*iters/s", "bytes_per_dof_iter": ITER_BYTES_PER_DOF,
                        "frac": ITER_BYTES_PER_DOF * n_global * Kd / td / 1e9 / PEAK_HBM_GBS,
                        "fusion": list(sd.fusion()), "x_update_every": sd.x_interval(),
                        "kernels": {k: {"ms": v, "bytes_per_row": SURVEY_BYTES[k], "GBps": SURVEY_BYTES[k] * n_global / v / 1e6,
                                        "frac": SURVEY_BYTES[k] * n_global / v / 1e6 / PEAK_HBM_GBS} for k, v in kd_ms.items()}}
        if xd_env is None:
            del os.environ["EC3D_XDEFER"]
        else:
            os.environ["EC3D_XDEFER"] = xd_env

    # Driver-timed figures for the reference's own system and for config 2's grid: the headline
    # handle is gone by now, each runs on a fresh handle, config.workload stays the 512^3 cube.
    side = {}
    if rank == 0 and not use_dist and not in_library and args.workload == "cube" and args.format == "dict" \
            and N == 512 and not args.no_side_workloads:
        for name in ("av", "av256", "cube256"):
            try:
                side[name] = side_workload(E, name, local_rank)
            except Exception as e:     # reporting only: the headline number does not depend on it
                side[name] = {"error": repr(e)}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / K
        value = n_global * K / elapsed
        rows = int(info.n)                       # rows one launch processes on this rank
        kernels, names, fmt_bytes, dom = stage_table(fusion_of, info, kernel_ms)
        kernel_ms = {k: v["ms"] for k, v in kernels.items()}
        tr = latest_traffic(N, args.format, args.workload, world)
        use_tr = bool(tr)
        achieved = fmt_bytes[dom] * rows / (kernel_ms[dom] * 1e-3) / 1e9
        traffic = tr["kernels"].get(dom, {}).get("hbm_bytes") if use_tr else None
        out = {
            "metric": "DOF*iters/s (fp64 BiCGSTAB-with-restart, 7-pt A-V operator)",
            "value": value, "unit": "DOF*iters/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "n": n_global, "grid": grid, "parallelism": parallelism,
                       "band_format": ("structured A-V form (1 class byte/row, U on the grid)"
                                       if args.workload in ("av", "av256") and info.tail_rows == 0 and info.dict_classes > 0
                                       else "dictionary (1 B/row + table)" if info.dict_classes > 0
                                       else "plain DIA"),
                       "workgroups": geom,
                       "bytes_per_dof_iter": {"survey_model": ITER_BYTES_PER_DOF,
                                              "this_format": sum(fmt_bytes[k] for k in kernel_ms)}},
            "iter_hbm_frac_survey_model": ITER_BYTES_PER_DOF * value / 1e9 / world / PEAK_HBM_GBS,
            "iter_hbm_frac": sum(fmt_bytes[k] for k in kernel_ms) * value / 1e9 / world / PEAK_HBM_GBS,
            "kernels": kernels,
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         # whose bytes `achieved` counts: "format" = what this storage format and these launches are built
                         # to move (class byte + LDS table instead of 56 B of coefficients per row, fused launches);
                         # "survey_8d" = SURVEY section 8d's plain-DIA five-kernel model (--format dia; the `iter_dia`
                         # sub-record carries the whole iteration on that model, `spmv_dia` the bare SpMV)
                         "byte_model": "format" if info.dict_classes > 0 else "survey_8d",
                         "traffic": traffic,
                         # not measured in this run: PMC counters need their own rocprofv3 passes
                         # (tools/profile_bench.sh); this is the committed profile of the same configuration
                         "traffic_source": (f"profiles/{tr['_file']} (rocprofv3 --pmc passes of this configuration, "
                                            f"FETCH_SIZE/WRITE_SIZE per the guide)" if traffic is not None else None),
                         "algorithmic_bytes_per_launch": fmt_bytes[dom] * rows,
                         "avg_launch_ms": kernel_ms[dom]},
        }
        if world == 1 and not use_dist and vplace[0]:
            out["config"]["vector_placement"] = {
                "candidate_us_per_iteration": [round(v, 1) for v in vplace[0]], "kept": vplace[1],
                "search_ms": round(vplace[2], 1),
                "note": "where the driver puts the work vectors is worth 2-3 % of the iteration at this size (each allocation "
                        "runs at its own time for as long as it lives), so at set-up -- outside the timed region -- the "
                        "library iterates a right-hand side of ones on up to EC3D_PLACE_VEC (6) allocations and keeps the "
                        "fastest (ec3d_get_vector_placement; profiles/r06_vector_placement.log)"}
        if x_every > 1:
            out["config"]["x_update_every"] = x_every
        if k4_spmv:
            out["config"]["k4_as_spmv"] = True
        if fusion_of.xg == 2:
            out["config"]["x_groups"] = "a launch of its own per group on the iteration's stream (k_x_group), timed with K4"
        if (in_library or use_dist) and verified:
            out["verified"] = verified
        if use_dist:
            out["transport"] = "rccl"
            out["rccl"] = rccl_record
            out["host"] = {"enqueue_ms_per_iteration": host_enqueue_ms, "api_calls_per_iteration": api_calls,
                           "note": "rank 0's host thread: time for ec3d_multi_iterate to ENQUEUE the timed iterations (launches, "
                                   "event records / waits, RCCL calls), which has to stay below ms_per_step"}
        if spmv_ms is not None:
            out["spmv"] = {"kernel": "k_spmv (y = A*x, 7 bands)", "ms": spmv_ms,
                           "survey_bytes_per_row": 72, "survey_GBps": 72 * rows / spmv_ms / 1e6,
                           "frac_of_peak_survey_model": 72 * rows / spmv_ms / 1e6 / PEAK_HBM_GBS,
                           "bytes_per_row": fmt_bytes["spmv"], "GBps": fmt_bytes["spmv"] * rows / spmv_ms / 1e6}
        if spmv_dia is not None:
            out["spmv_dia"] = spmv_dia
        if iter_dia is not None:
            out["iter_dia"] = iter_dia
        for name, rec in side.items():
            out[name] = rec
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_grid, args.cpu_iters)
            except Exception as e:  # the baseline is reporting only; never fail the GPU number on it
                out["cpu_baseline"] = {"value": None, "unit": "DOF*iters/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
            try:    # the labelled all-cores column: context, never credit
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.cpu_grid, args.cpu_iters)
            except Exception as e:
                out["cpu_baseline_all_cores"] = {"value": None, "unit": "DOF*iters/s", "cores": 0, "kind": "port (OpenMP)",
                                                 "sample": f"failed: {e!r}"}
    if in_library:
        s.close()
    elif use_dist:
        s.close()              # the communicators go first
        dist.barrier()
        dist.destroy_process_group()
        # the other transport as a sub-record: a fresh child process running the in-library form on the same devices, now
        # that this job's ranks are done with them (rank 0 only; the other ranks have nothing left to do)
        if rank == 0 and world > 1 and not args.rehearse and not args.no_in_library:
            out["in_library"] = run_in_library_child(args, list(range(world)), args.child_timeout)
    else:
        s.close()
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
