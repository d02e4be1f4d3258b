#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 counter_collection.csv files."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-40:]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            acc[name]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
want = [k for k in acc if "skinny" in k or "attn_softmax_context" in k or "gemm_f32_kernel<128, 128, 16, true, true>" in k]
for k in sorted(want):
    print(k, "n=%d" % len(acc[k]["_dur_ns"]))
    for c, v in sorted(acc[k].items()):
        print(f"   {c:34s} {sum(v)/len(v):16.1f}")
