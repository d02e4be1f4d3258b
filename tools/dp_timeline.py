#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 kernel_trace.csv: kernels in start order with
queue, duration and the idle gap on the whole device before each (kernels longer than 30 us and every gap > 5 us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][-48:]
# find the adam kernels: a step ends with the second adam launch
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
if len(adam) < 8:
    sys.exit("not enough steps in the trace")
lo, hi = adam[-7] + 1, adam[-5]            # one full step: after the adam pair of step n-1 .. the adam pair of step n
step = rows[lo:hi + 1]
t0 = int(step[0]["Start_Timestamp"])
print(f"step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us, {len(step)} kernels")
busy_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - busy_end) / 1e3
    if (e - s) > 25000 or gap > 3 or "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower():
        print(f"t={(s - t0) / 1e3:8.1f}  q{r['Queue_Id']:>2s}  dur {(e - s) / 1e3:8.1f}  idle-before {gap:6.1f}  {nm(r)}")
    busy_end = max(busy_end, e)
