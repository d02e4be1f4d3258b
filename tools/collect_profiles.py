#!/usr/bin/env python3
"""Copy what `tools/profile_round.sh <round> <workloads>` left under gpurun_out/<dir>/ into profiles/ under the round's names
and merge the per-workload entries of attn_in_step.json / attn_traffic.json (the box starts from the repository's copies):
    python tools/collect_profiles.py gpurun_out/r04g r04 config2 [config4 ...]"""
import json
import os
import shutil
import sys

src, rnd, workloads = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, "profiles")
copied = []


def cp(name, dst):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copyfile(p, os.path.join(prof, dst))
        copied.append(dst)


for w in workloads:
    for name in (f"bench_{w}.json", f"kernel_stats_{w}.csv", f"kernel_stats_{w}_step_only.csv", f"phase_times_{w}.txt",
                 f"persist_dec_trace_{w}.txt", f"persist_dec_bwd_trace_{w}.txt", f"step_timeline_{w}.txt"):
        cp(name, f"{rnd}_{name}")
    cp(f"pmc_fetch_attn_{w}.csv", f"{rnd}_pmc_FETCH_SIZE_attn_fwd_{w}.csv")
    cp(f"pmc_write_attn_{w}.csv", f"{rnd}_pmc_WRITE_SIZE_attn_fwd_{w}.csv")
if "config2" in workloads:
    for name in ("bench_config2_same_box.json", "bench_config2_force_dp.json", "bench_config2_force_dp_noreserve.json", "x3_fixed_cost.txt"):
        cp(name, f"{rnd}_{name}")
for name in ("attn_in_step.json", "attn_traffic.json"):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        continue
    base = json.load(open(os.path.join(prof, name)))
    new = json.load(open(p))
    for k, v in new.items():
        if any(k == w or k.startswith(w + ":") for w in workloads):
            base[k] = v
    json.dump(base, open(os.path.join(prof, name), "w"), indent=1)
    copied.append(name)
print("\n".join(copied))
