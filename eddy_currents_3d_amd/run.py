"""Run a VoxCad ``.vxc`` model the way the reference program does, on one MI355X.

    python -m eddy_currents_3d_amd.run model.vxc [--steps N] [--out DIR] [--device D]
    python -m torch.distributed.run --nproc-per-node G -m eddy_currents_3d_amd.run model.vxc ...   (G GPUs)

Reads the file (eddy_currents_3d_amd/vxc.py), assembles the A-V system on the device, and runs the
reference's time loop (eddy_currents_3d_amd/host.py) with the fields resident in HBM; ``field_N.vtk`` files
and ``src_N.vtk`` files go to ``--out`` (default: the ``dir=`` name of the model's solver line, as the
reference does).
"""
from __future__ import annotations

import argparse
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m eddy_currents_3d_amd.run", description=__doc__.split("\n\n")[0])
    ap.add_argument("model", help="path of the .vxc file")
    ap.add_argument("--steps", type=int, default=None, help="stop after this many time steps (default: the model's stop time)")
    ap.add_argument("--out", default=None, help="directory for field_N.vtk (default: the model's dir= name)")
    ap.add_argument("--no-output", action="store_true", help="do not write VTK files")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--refine", type=int, nargs=3, metavar=("FX", "FY", "FZ"), default=None,
                    help="split every cell FX x FY x FZ times first (same physical size, finer grid)")
    a = ap.parse_args(argv)

    from . import EC3DSolver, host, vxc
    model = vxc.read_vxc(a.model)
    if a.refine:
        model = vxc.refine(model, *a.refine)
    t = vxc.domain_tables(model)
    sdz, sdy, sdx = model.vox.shape
    out_dir = None if a.no_output else (a.out or str(t["directory"]).upper())
    print(f"{a.model}: grid {sdx}x{sdy}x{sdz}, {t['ncells0']} conducting cells, dt={t['dt']:g} stop={t['time']:g} "
          f"tol={t['tol']:g} itmax={t['itmax']}", flush=True)
    t0 = time.perf_counter()

    def on_step(k, s, info):
        print(f"step {k:4d}  T={info['T']:.6g}  iter={info['iter']}"
              + (f"  -> field_{info['output']}.vtk" if "output" in info and out_dir else ""), flush=True)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:   # one process per GPU: z-slabs, halo exchange and reductions over RCCL
        import torch
        import torch.distributed as dist
        rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        try:
            log = host.run_slabs(model, rank, world, device=local, steps=a.steps, out_dir=out_dir,
                                 on_step=on_step if rank == 0 else None)
        finally:
            dist.destroy_process_group()
        if rank != 0:
            return 0
        n = 3 * model.vox.size + t["ncells0"]
    else:
        with EC3DSolver(device=a.device) as s:
            log = host.run(model, s, steps=a.steps, out_dir=out_dir, on_step=on_step)
            n = s.n
    wall = time.perf_counter() - t0
    its = sum(i["iter"] for i in log)
    print(f"{len(log)} steps, {its} solver iterations, n={n}, {wall:.2f} s wall "
          f"({n * its / wall:.3e} DOF*iters/s including assembly, source update and output)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
