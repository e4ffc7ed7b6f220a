#!/usr/bin/env python3
"""Condense rocprofv3 output (tools/profile_bench.sh) into the small files kept under profiles/:
   <tag>_kernel_stats.csv   per-kernel calls / total / average duration (from --kernel-trace --stats)
   <tag>_pmc_traffic.json   HBM bytes per launch of each kernel from FETCH_SIZE / WRITE_SIZE,
                            corrected as MI355X_MICROARCH.md §HBM prescribes and calibrated on K2
"""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, tag, grid = sys.argv[1], sys.argv[2], int(sys.argv[3])
fmt = sys.argv[4] if len(sys.argv) > 4 else "dict"
workload = sys.argv[5] if len(sys.argv) > 5 else "cube"
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(repo, "gpurun_out", "profiles_" + tag)
os.makedirs(prof, exist_ok=True)


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name


# ---- kernel stats
stats = find("trace", "*kernel_stats.csv")
rows = []
if stats:
    with open(stats) as f:
        for r in csv.DictReader(f):
            rows.append(r)
    with open(os.path.join(prof, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"],
                        r["MaxNs"], r["Percentage"]])
        print(open(os.path.join(prof, f"{tag}_kernel_stats.csv")).read())

# ---- PMC
def pmc(sub, counter):
    fn = find(sub, "*counter_collection.csv")
    acc = defaultdict(list)
    if not fn:
        return acc
    with open(fn) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc

fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
n = grid ** 3
try:  # rows one launch processes: what bench.py itself reported under the profiler
    with open(os.path.join(out, "bench_trace.json")) as f:
        n = int(json.loads(f.read().strip().splitlines()[-1])["config"]["n"])
except (OSError, ValueError, KeyError, IndexError):
    pass
res = {"tag": tag, "grid": grid, "n_gpus": 1, "format": fmt, "workload": workload, "rows": n, "units_note":
       "FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (gfx950 reports 1/2 of a "
       "16-B-per-lane stream); per launch = mean over the launches of the profiled run", "kernels": {}}
def stage_of(k, have_k4d):
    if k.startswith("k23_"):   # K2 inside K3: bench.py books it under k3
        return "k3"
    if k.startswith("k51_"):   # K5 inside the next K1: booked under k5
        return "k5"
    if k.startswith("k4d_") or k.startswith("k4s_"):   # K4 with the X update deferred (vector or SpMV form): launches without X and applying launches, pooled --
        return "k4"            # the mean over the launches of the run is the mean bench.py's kernels.k4.ms is
    if k.startswith("k4_") and have_k4d:
        return "k4_classic"    # (the lone launches of iteration 1 of a set-up pass)
    m = re.match(r"(k[1-5])_", k)
    return m.group(1) if m else k


names = sorted(set(fetch) | set(write))
have_k4d = any(k.startswith("k4d_") or k.startswith("k4s_") for k in names)
pool_f, pool_w, pool_n = defaultdict(list), defaultdict(list), defaultdict(list)
for k in names:
    kk = stage_of(k, have_k4d)
    pool_f[kk] += fetch.get(k, [])
    pool_w[kk] += write.get(k, [])
    pool_n[kk].append(f"{k} x{max(len(fetch.get(k, [])), len(write.get(k, [])))}")
for kk in pool_n:
    fv = sum(pool_f[kk]) / len(pool_f[kk]) if pool_f[kk] else None
    wv = sum(pool_w[kk]) / len(pool_w[kk]) if pool_w[kk] else None
    res["kernels"][kk] = {"rocprof_name": ", ".join(pool_n[kk]), "fetch_kib_raw": fv, "write_kib_raw": wv,
                         "hbm_read_bytes": None if fv is None else 2 * fv * 1024,
                         "hbm_write_bytes": None if wv is None else wv * 1024}
for k, v in res["kernels"].items():
    if v["hbm_read_bytes"] is not None and v["hbm_write_bytes"] is not None:
        v["hbm_bytes"] = v["hbm_read_bytes"] + v["hbm_write_bytes"]
        v["bytes_per_row"] = v["hbm_bytes"] / n
with open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
