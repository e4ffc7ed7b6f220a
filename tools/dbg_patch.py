import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import eddy_currents_3d_amd as E
from oracle import oracle as O
for dims in [(128, 32, 16), (128, 16, 12), (256, 8, 9), (128, 32, 24)]:
    sdx, sdy, sdz = dims
    valA, irow, jcol = O.poisson_csr(sdx, sdy, sdz)
    n = sdx * sdy * sdz
    x = np.random.Generator(np.random.PCG64(21)).standard_normal(n)
    yo = O.spmv_csr(valA, irow, jcol, x)
    for rep in range(3):
        with E.EC3DSolver() as s:
            s.assemble_poisson(sdx, sdy, sdz)
            g = s.geometry(1)
            y = s.spmv(x)
            bad = np.flatnonzero(y != yo)
            print(dims, "patch", g.patch_x, "nblk", g.nblk, "tpp", g.zm_tpp, "pps", g.zm_pps, "bad rows", bad.size,
                  (bad.min(), bad.max(), (bad // (sdx * sdy)).min(), (bad // (sdx*sdy)).max()) if bad.size else "", flush=True)
