#!/bin/bash
set -e
python3 tools/air_box.py 64 64 8 tiny
python3 tools/air_box.py 128 128 16 small
python3 tools/air_box.py 256 256 32 mid
python3 tools/air_box.py 306 306 72 full
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import eddy_currents_3d_amd as E
for N in (64, 128, 256):
    with E.EC3DSolver() as s:
        s.assemble_poisson(N, N, N)
        s.upload("B", np.ones(N**3)); s.upload("X", np.zeros(N**3))
        s.iterate_begin(); s.iterate(1, 5); s.synchronize()
        ms = s.iterate(6, 40, per_kernel=True)
        print(f"cube {N}: " + " ".join(f"{k}={1e3*v:7.1f}" for k, v in ms.items()), f"wg={s.geometry(0).nblk}/{s.geometry(1).nblk}", flush=True)
PY
