#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter CSVs into per-launch HBM-side traffic for one kernel.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <workload> [out.json] [source.hip] [key]

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte requests as
64 bytes for wide coalesced streaming reads (MI355X_MICROARCH.md, section HBM), so the read side
is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  The counters sit on the
L2's memory side: Infinity-Cache hits are included, so this is "bytes that left L2", the figure to
hold against the algorithmic bytes (re-reads show up here whether HBM or the MALL served them).

The record names the kernel exactly as the profiler saw it (template arguments included), its grid
and the sha256 of csrc/attention.hip at collection time; bench.py reports ``traffic`` only when that
hash matches the source it runs."""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(d, counter, kernel):
    vals, names, grids = [], set(), set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
                vals.append(float(row["Counter_Value"]))
                names.add(row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""))
                grids.add(int(row["Grid_Size"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for kernel *{kernel}* under {d}")
    return sum(vals) / len(vals), len(vals), sorted(names), sorted(grids)


def main():
    fd, wd, kernel, workload = sys.argv[1:5]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "attn_traffic.json")
    fetch_kib, n1, names, grids = per_launch(fd, "FETCH_SIZE", kernel)
    write_kib, n2, _, _ = per_launch(wd, "WRITE_SIZE", kernel)
    src = os.path.join(ROOT, "multi-modal-qg_amd", "csrc", sys.argv[6] if len(sys.argv) > 6 else "attention.hip")
    key = sys.argv[7] if len(sys.argv) > 7 else workload
    rec = {"kernel": ", ".join(names), "grid": grids, "workgroups": [g // (512 if "persist" in kernel else 256) for g in grids],
           "source_sha": hashlib.sha256(open(src, "rb").read()).hexdigest()[:16],
           "launches_fetch": n1, "launches_write": n2,
           "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
           "hbm_bytes_per_launch": int(2 * fetch_kib * 1024 + write_kib * 1024),
           "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B), WRITE_SIZE x1; L2 memory-side "
                         "counters, Infinity-Cache hits included"}
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[key] = rec
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
