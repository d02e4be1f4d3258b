#!/usr/bin/env python3
"""Every kernel of one steady-state step from a rocprofv3 kernel_trace.csv, in start order: start, queue, duration."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# (a step starts with the batch repack; the optimizer launches are no marker: their number depends on the arrangement)
pack = [i for i, r in enumerate(rows) if "pack_batch_kernel" in r["Kernel_Name"]]
step = rows[pack[-3]:pack[-2]]
t0 = int(step[0]["Start_Timestamp"])
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    print(f"{(s - t0) / 1e3:8.1f} q{r['Queue_Id']:>2s} {(e - s) / 1e3:7.1f} {n} grid={r.get('Grid_Size') or r.get('Grid_Size_X', '?')}")
