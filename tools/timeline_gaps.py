#!/usr/bin/env python3
"""How much of a rank's iteration is the GPU NOT running one of this library's kernels?  From a rocprofv3 --kernel-trace
CSV: the kernels of the steady part of the run (the last `frac` of the trace), merged into busy intervals of (a) the
library's kernels alone, (b) every kernel (RCCL's included); prints per iteration: wall, busy time of each, and the
histogram of the gaps between the library's kernels -- the room an independent kernel on another stream could fill.
usage: timeline_gaps.py <kernel_trace.csv> <iterations per K4 count: name of a once-per-iteration kernel prefix>"""
import csv, sys
fn, once = sys.argv[1], sys.argv[2]
rows = []
with open(fn) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = len(rows)
rows = rows[n // 2:]                     # steady part
t0, t1 = rows[0][0], max(r[1] for r in rows)
its = sum(1 for r in rows if once in r[2])


def busy(sel):
    iv = sorted((a, b) for a, b, k in rows if sel(k))
    tot, gaps, cur_a, cur_b = 0, [], None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
                gaps.append(a - cur_b)
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    tot += cur_b - cur_a
    return tot, gaps


is_rccl = lambda k: "nccl" in k.lower() or "rccl" in k.lower()
lib, gaps = busy(lambda k: not is_rccl(k))
allk, _ = busy(lambda k: True)
rc, _ = busy(is_rccl)
print(f"{its} iterations over {(t1 - t0) / 1e3:.0f} us: {(t1 - t0) / its / 1e3:.1f} us per iteration; library kernels busy "
      f"{lib / its / 1e3:.1f} us, any kernel busy {allk / its / 1e3:.1f} us, RCCL kernels busy {rc / its / 1e3:.1f} us per iteration")
edges = [0, 2, 5, 10, 20, 40, 80, 10 ** 9]
for lo, hi in zip(edges[:-1], edges[1:]):
    g = [x for x in gaps if lo * 1e3 <= x < hi * 1e3]
    print(f"  gaps {lo:>3}-{hi if hi < 10 ** 9 else 'inf':>3} us between library kernels: {len(g) / its:6.2f} per iteration, {sum(g) / its / 1e3:6.1f} us per iteration")
names = {}
for a, b, k in rows:
    s = k.split("(")[0][:70]
    c = names.setdefault(s, [0, 0])
    c[0] += 1
    c[1] += b - a
for s, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {c / its:5.2f} x {t / c / 1e3:7.1f} us  {s}")
