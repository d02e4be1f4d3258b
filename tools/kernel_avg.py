#!/usr/bin/env python3
"""Average duration per kernel name from a rocprofv3 kernel-trace CSV (skipping the first `skip` launches of each name):
    python tools/kernel_avg.py <dir or *_kernel_trace.csv> [skip] [name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
filt = sys.argv[3:]
if os.path.isdir(src):
    src = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True))[0]
d = defaultdict(list)
with open(src) as f:
    for r in csv.DictReader(f):
        d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows = []
for k, v in d.items():
    v.sort()
    v = v[skip:] if len(v) > skip else v
    if filt and not any(x in k for x in filt):
        continue
    rows.append((sum(e - s for s, e in v) / len(v) / 1e3, len(v), k))
for avg, n, k in sorted(rows, reverse=True)[:14]:
    print(f"{avg:9.1f} us  x{n:4d}  {k[:100]}")
