#!/usr/bin/env python3
"""End-to-end drop-in run on the GPU box: the UNMODIFIED reference program (oracle/_ref/EC3D_dropin:
its own .vxc ingest, assembly, coil motion, RHS build, time loop) with every solve going through the
`sprsbcgstabwr_` exported by libec3d_hip.so, on a shipped input refined by integer factors
(BASELINE config 3 style).  With --reference the same input is also run through the pure reference
(oracle/_ref/EC3D_capture) for `--ref-steps` steps to time the CPU solver on the same box.

Lives under tests/ because it runs the reference's own binaries (oracle/_ref): checker infrastructure, not
product.  Not collected by pytest (run it by hand on the GPU box).

usage: dropin_scaled.py <compare_to_Elmer|ec_src_move_hole|LIM> fx fy fz steps [--reference --ref-steps K]
"""
import argparse, os, re, subprocess, sys, tempfile, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
from oracle import make_goldens as G
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("case"); ap.add_argument("fx", type=int); ap.add_argument("fy", type=int); ap.add_argument("fz", type=int)
ap.add_argument("steps", type=int)
ap.add_argument("--reference", action="store_true"); ap.add_argument("--ref-steps", type=int, default=1)
ap.add_argument("--tol", default=None, help="override the input's solver tolerance, e.g. 1e-8 (palette 'solver tol=')")
ap.add_argument("--compare", action="store_true",
                help="with --reference: capture every call's x on both sides and print the relative difference")
a = ap.parse_args()

from eddy_currents_3d_amd import vxc
g = np.load(os.path.join(REPO, "tests", "golden", f"g4_{a.case}.npz"))
# no VTK output during the timed run: JUMP beyond the stop time (src/vxc2data.f90:191-195, EC3D.f90:143-144)
names = [re.sub(r"(tran\b.*)", r"\1 jump=1000", str(s), flags=re.I) if re.search(r"\btran\b", str(s), re.I)
         else str(s) for s in g["names"]]
if a.tol:
    names = [re.sub(r"\btol=\S+", "tol=" + a.tol, n) if re.search(r"\bsolver\b", n, re.I) else n for n in names]
base = vxc.VxcModel(g["vox"], names, float(str(g["lattice_dim"])), tuple(float(x) for x in g["adj"]))
model = vxc.refine(base, a.fx, a.fy, a.fz)   # keeps the physical size: cell size / factor per axis
vox = model.vox
sdz, sdy, sdx = vox.shape
print(f"{a.case} x({a.fx},{a.fy},{a.fz}): grid {sdx}x{sdy}x{sdz} = {vox.size} cells", flush=True)


def read_x(cap_dir):
    """x_out of every captured call (layout: oracle/capture_interposer.c)."""
    xs = []
    for fn in sorted(os.listdir(cap_dir)):
        with open(os.path.join(cap_dir, fn), "rb") as f:
            n, nnz, itmax, it = np.fromfile(f, np.int64, 4)
            f.seek(16 + 4 * (n + 1) + (12 * nnz if nnz else 0) + 8 * 2 * n, 1)
            xs.append(np.fromfile(f, np.float64, n))
    return xs


def run(exe, steps, env_extra, capture=False):
    t0 = time.time()
    td = tempfile.mkdtemp(prefix="ec3d_dropin_")
    vxc.write_vxc(os.path.join(td, "in.vxc"), model, compression="ASCII_READABLE")
    with open(os.path.join(td, "del"), "w") as f:
        f.write("#!/bin/sh\nexit 0\n")
    os.chmod(os.path.join(td, "del"), 0o755)
    env = dict(os.environ, PATH=td + ":" + os.environ["PATH"], EC3D_CAPTURE_MAX_CALLS=str(steps), **env_extra)
    env.pop("EC3D_CAPTURE_DIR", None)
    if capture:
        env["EC3D_CAPTURE_DIR"] = os.path.join(td, "cap")
        os.mkdir(env["EC3D_CAPTURE_DIR"])
    import threading
    done = threading.Event()

    def heartbeat():  # long CPU runs must not look hung to the job runner
        while not done.wait(60.0):
            print(f"  ... {os.path.basename(exe)} running, {time.time() - t0:.0f} s", flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    p = subprocess.run([exe], cwd=td, env=env, preexec_fn=O._unlimit_stack, stdout=subprocess.DEVNULL,
                       stderr=subprocess.PIPE)
    done.set()
    calls = re.findall(r"\[capture\] call (\d+) n=(\d+) nnz=(\d+) iter=(\d+) t=([\d.]+)s", p.stderr.decode())
    xs = read_x(env["EC3D_CAPTURE_DIR"]) if capture else None
    subprocess.run(["rm", "-rf", td])
    if capture:
        return calls, time.time() - t0, p.stderr.decode()[-500:], xs
    return calls, time.time() - t0, p.stderr.decode()[-500:]


lib = os.path.join(REPO, "eddy_currents_3d_amd", "libec3d_hip.so")
calls, wall, err = run(os.path.join(REPO, "oracle", "_ref", "EC3D_dropin"), a.steps, {"EC3D_HIP_LIB": lib})
if not calls:
    print("no solver call captured:", err); sys.exit(1)
n, nnz = int(calls[0][1]), int(calls[0][2])
its = [int(c[3]) for c in calls]; ts = [float(c[4]) for c in calls]
print(f"GPU drop-in : n={n} nnz={nnz} steps={len(calls)} iters={its}")
print(f"              solve call s/step={['%.3f' % t for t in ts]}  (first includes CSR->device conversion)")
print(f"              whole program wall {wall:.1f} s (ingest + reference assembly + {len(calls)} steps)")
rate = [n * i / t for i, t in zip(its[1:], ts[1:])] or [n * its[0] / ts[0]]
print(f"              DOF*iters/s per call incl. H2D/D2H: {np.mean(rate):.3e}")
if a.reference and a.compare:
    k = a.ref_steps
    gc, _, _, gx = run(os.path.join(REPO, "oracle", "_ref", "EC3D_dropin"), k, {"EC3D_HIP_LIB": lib}, capture=True)
    rc, wall, err, rx = run(os.path.join(REPO, "oracle", "_ref", "EC3D_capture"), k, {}, capture=True)
    for q in range(min(len(gx), len(rx))):
        rel = np.linalg.norm(gx[q] - rx[q]) / np.linalg.norm(rx[q])
        print(f"step {q}: iter gpu {gc[q][3]} / reference {rc[q][3]}  ||x_gpu - x_ref|| / ||x_ref|| = {rel:.3e}"
              f"  (reference solve {float(rc[q][4]):.1f} s on one core, gpu call {float(gc[q][4]):.3f} s)")
elif a.reference:
    calls, wall, err = run(os.path.join(REPO, "oracle", "_ref", "EC3D_capture"), a.ref_steps, {})
    its = [int(c[3]) for c in calls]; ts = [float(c[4]) for c in calls]
    print(f"CPU reference: steps={len(calls)} iters={its} solve s/step={['%.2f' % t for t in ts]} "
          f"-> {np.mean([n * i / t for i, t in zip(its, ts)]):.3e} DOF*iters/s (1 core); program wall {wall:.1f} s")
