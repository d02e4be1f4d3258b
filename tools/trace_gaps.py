#!/usr/bin/env python3
"""Per-kernel durations and idle gaps from a rocprofv3 kernel_trace.csv (graph replay run)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# take the last third (steady-state replays)
rows = rows[int(n * 0.6):]
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
busy = 0; span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][-60:]
    dur[name].append(e - s)
    if prev_end is not None:
        gap_after[name].append(s - prev_end)
    prev_end = max(prev_end or 0, e)
    busy += e - s
print(f"span {span/1e3:.1f} us, sum of kernel durations {busy/1e3:.1f} us ({100*busy/span:.1f}% if serial)")
print(f"{'kernel':62s} {'n':>6s} {'avg_us':>8s} {'gap_before_us':>14s}")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:18]:
    g = gap_after.get(k, [0])
    print(f"{k:62s} {len(v):6d} {sum(v)/len(v)/1e3:8.2f} {sum(g)/len(g)/1e3:14.2f}")
