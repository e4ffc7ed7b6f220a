#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in a .hip source (no GPU needed):
   python tools/resource_usage.py [source.hip] [filter]
Compiles with the library's flags plus -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
ScratchSize > 0 means the kernel spills."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.join(REPO, "eddy_currents_3d_amd", "csrc", "ec3d_kernels.hip")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-c", src, "-o",
       "/tmp/_ru.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|TotalSGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
try:
    names = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
except OSError:
    names = [r["name"] for r in rows]
print(f"{'kernel':70s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'occ':>4s} {'LDS':>6s} {'Sspill':>6s} {'Vspill':>6s}")
for r, nm in zip(rows, names):
    nm = re.sub(r"\(.*", "", nm).replace("void ", "")
    if flt and flt not in nm:
        continue
    print(f"{nm[:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize','?'):>8s} {r.get('Occupancy','?'):>4s} {r.get('LDS Size','?'):>6s} {r.get('SGPRs Spill','?'):>6s} {r.get('VGPRs Spill','?'):>6s}")
