#!/usr/bin/env python3
"""A-V slabs on plan 5 (K2 / K5 boundary tiles first AND K1 / K3 interior first) against plan 2 on the same slabs: the LIM
geometry resampled to a small grid, `world` slabs on one GPU inside the library.  Prints plan, iterations, the relative
distance of the two solutions and whether the interior and boundary launches of K1 / K3 visit every owned tile once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
from eddy_currents_3d_amd import vxc

dims = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 32, 48)
world = int(sys.argv[4]) if len(sys.argv) > 4 else 2
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g4_LIM.npz"))
model = vxc.resample(vxc.VxcModel(g["vox"], [str(x) for x in g["names"]], float(str(g["lattice_dim"])),
                                  tuple(float(x) for x in g["adj"])), *dims)
t = vxc.domain_tables(model)
res = {}
for plan in (2, 5):
    os.environ["EC3D_SLAB_PLAN"] = str(plan)
    with E.EC3DMulti(world, devices=[0] * world) as m:
        m.assemble(t["geoPHYS"], t["geoPHYS_C"], t["valPHYS"], t["BND"], t["delta"], t["dt"])
        n = m.n
        b = m.spmv(np.random.Generator(np.random.PCG64(3)).standard_normal(n))     # a right-hand side in the range of A
        m.upload("B", b); m.upload("X", np.zeros(n))
        m.iterate_begin(); m.iterate(1, 1); m.synchronize()
        ap = m.download("AP")
        x, it = m.solve(b, np.zeros(n), 1e-8, 5000)
        rel, bn = m.true_residual()
        cover = []
        for r in range(world):
            v = m.slab(r)[0]
            whole = np.sort(v.visit_order(1)[1])
            if v.can_overlap():
                parts = np.sort(np.concatenate([v.visit_order(3)[1], v.visit_order(4)[1]]))
                ok = bool(np.array_equal(parts, whole))
                cover.append(ok)
                if not ok:
                    sw, sp = set(whole.tolist()), set(parts.tolist())
                    print(f"   slab {r}: whole {len(whole)} tiles ({len(sw)} distinct), parts {len(parts)} ({len(sp)} distinct); missing "
                          f"{sorted(sw - sp)[:12]} extra {sorted(sp - sw)[:12]}")
            else:
                cover.append(None)
        res[plan] = (x, it, ap)
        print(f"plan asked {plan}: runs {m.plan()}, n={n}, iter {it}, true residual {rel:.2e}, K1/K3 interior+boundary cover the owned tiles once: {cover}", flush=True)
d = np.linalg.norm(res[5][0] - res[2][0]) / np.linalg.norm(res[2][0])
print(f"plan 5 against plan 2: iterations {res[5][1]} / {res[2][1]}, relative distance of x {d:.2e}; AP = A P of the first iteration "
      f"the same bits: {bool(np.array_equal(res[5][2], res[2][2]))} (|AP| = {np.linalg.norm(res[2][2]):.6e})")
