#!/usr/bin/env python3
"""In-step duration of the decoder-attention forward kernel from a rocprofv3 kernel-trace of
`bench.py --kernel-iters 0` (the CSV then holds only the launches the training steps make: no back-to-back or
rotating probe loops), written to profiles/attn_in_step.json with the hash of the kernel source.

    python tools/attn_in_step.py <kernel_trace.csv> <workload> [out.json] [warmup_launches_to_skip]

bench.py reports roofline.frac from this record (in-step: what the step actually runs) when its hash matches the
attention.hip it loads, next to the live back-to-back and beyond-Infinity-Cache figures."""
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path, workload = sys.argv[1:3]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "attn_in_step.json")
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    all_rows = list(csv.DictReader(open(path)))
    rows = [r for r in all_rows if "attn_softmax_context_fwd_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[skip:]
    src = os.path.join(ROOT, "multi-modal-qg_amd", "csrc", "attention.hip")
    if not rows:
        # the decoder's forward loop ran as one persistent launch (csrc/persist_dec.hip): the attention is a phase of it
        rows = [r for r in all_rows if "decoder_persist_fwd_kernel" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        rows = rows[3:]                                     # the warm-up steps
        src = os.path.join(ROOT, "multi-modal-qg_amd", "csrc", "persist_dec.hip")
    if not rows:
        raise SystemExit("no attention forward launches and no persistent decoder launches in " + path)
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    rec = {"kernel": rows[0]["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""),
           "launches": len(d), "us_per_launch": round(sum(d) / len(d) / 1e3, 3), "min_us": min(d) / 1e3, "max_us": max(d) / 1e3,
           "grid": sorted({int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0) for r in rows}),
           "source_sha": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           "from": "rocprofv3 --kernel-trace of `bench.py --workload %s --kernel-iters 0`: step launches only" % workload}
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[workload] = rec
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
