#!/usr/bin/env python3
"""Per-kernel mean of every counter in the rocprofv3 --pmc output under a directory:  pmc_avg.py <dir> [name filter]"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
            name = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if flt in k:
        print(k, {c: round(sum(v) / len(v), 1) for c, v in sorted(acc[k].items())}, "launches", max(len(v) for v in acc[k].values()))
