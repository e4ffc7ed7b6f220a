import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import eddy_currents_3d_amd as E
from oracle import oracle as O
os.environ["EC3D_PITCH"] = "2"
os.environ["EC3D_SAV_QUAD"] = "1"
src = open(os.path.join(os.path.dirname(__file__), "..", "tests", "test_gpu_formats_dist.py")).read()
ns = {}
exec(src[src.index("def synthetic_av"):src.index('@pytest.mark.parametrize("fuse", ["0", "2"])')], {"np": np}, ns)
cases = [((64, 22, 12), "32", (9, 52, 4, 19, 3, 9), (28, 36, 9, 14)), ((96, 21, 10), "48", (5, 90, 3, 18, 2, 8), (40, 60, 8, 13)),
         ((102, 23, 9), None, (7, 95, 4, 20, 2, 7), (50, 56, 10, 14))]
for dims, px, block, hole in cases:
    if px: os.environ["EC3D_SAV_PATCH_PX"] = px
    else: os.environ.pop("EC3D_SAV_PATCH_PX", None)
    geo, geoC, valPHYS, BND, delta, dt = ns["synthetic_av"](*dims, block, hole)
    m = O.gen_sparse_matrix(geo, geoC, valPHYS, BND, delta, dt)
    x = np.random.Generator(np.random.PCG64(606)).standard_normal(m["n"])
    yo = O.spmv_csr(m["valA"], m["irow"], m["jcol"], x)
    with E.EC3DSolver() as s:
        s.assemble(geo, geoC, valPHYS, BND, delta, dt)
        y = s.spmv(x)
        bad = np.flatnonzero(y != yo)
        print(dims, px, "bad rows", bad.size, bad[:10], flush=True)
os.environ.pop("EC3D_SAV_PATCH_PX", None)
for name in ("g2_conducting_hole_16x15x14", "g3_moving_coil_18x16x12"):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))
    x = np.random.Generator(np.random.PCG64(31)).standard_normal(len(g["irow"]) - 1)
    with E.EC3DSolver() as s:
        s.assemble(g["geoPHYS"], g["geoPHYS_C"], g["valPHYS"], g["BND"], g["delta"], float(g["dt"]))
        y = s.spmv(x)
        print(name, "bad rows", int(np.count_nonzero(y != O.spmv_csr(g["valA"], g["irow"], g["jcol"], x))), flush=True)
