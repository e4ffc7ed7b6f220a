#!/usr/bin/env python3
"""tests/golden/flang_list_directed.json: doubles and the text flang's list-directed `print *, x` writes for them --
the format of the one line the reference's solver prints (`print*, norm2(R)`, src/solvers.f90:27) when the reference is
built with the toolchain of this image (oracle/Makefile: amdflang).  A scratch Fortran program of OURS (three lines: read
an array, print each element) is compiled with amdflang and run; nothing of the reference is involved.  Run here (the
GPU box uses the committed fixture)."""
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261004)
vals = np.concatenate([
    np.array([0.58139877942062257, 16.270496299768709, 1.2345678901234567e-5, 9.87654321e-3, 0.1, 0.0999, 123456789.123,
              1e16, 1.234e17, 1e-300, 0.0, 5e2, 0.05, 0.011, 0.00999, 0.01, 1e15, 9.9e15, 1.2345e16, 1234567890123456.7,
              12345678901234567.0, -0.5, -3.25e-7, 1.5, 408.47320541999864, 0.08796511733349222, 9.510229811957995]),
    rng.random(150) * 10.0 ** rng.integers(-12, 18, 150), rng.standard_normal(40)])
with tempfile.TemporaryDirectory() as td:
    vals.astype("<f8").tofile(os.path.join(td, "v.bin"))
    with open(os.path.join(td, "p.f90"), "w") as f:
        f.write(f"program p\n  real(8) :: v({len(vals)})\n  integer :: i\n"
                f"  open(10, file='{td}/v.bin', access='stream', form='unformatted')\n  read(10) v\n"
                f"  do i = 1, {len(vals)}\n    print *, v(i)\n  end do\nend program\n")
    subprocess.run(["/opt/rocm/bin/amdflang", os.path.join(td, "p.f90"), "-o", os.path.join(td, "p")], check=True,
                   capture_output=True, timeout=120)
    out = subprocess.run([os.path.join(td, "p")], capture_output=True, text=True, check=True, timeout=60).stdout.split("\n")
out = out[:len(vals)]
with open(os.path.join(HERE, "..", "tests", "golden", "flang_list_directed.json"), "w") as f:
    json.dump({"values_hex": [float(v).hex() for v in vals], "text": out}, f)
print(len(vals), "values")
