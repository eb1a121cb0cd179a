#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof_rNN) into the tracked profiles/ directory.

    python tools/summarize_prof.py gpurun_out/prof_r01 r01

Copies <prefix>_kernel_stats.csv (the `--kernel-trace --stats` summary) and writes
rNN_pmc_summary.json with per-kernel FETCH_SIZE / WRITE_SIZE (separate --pmc passes).
HBM bytes per launch follow MI355X_MICROARCH.md section HBM: counters are in KiB, and on
gfx950 FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads,
so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)
for f in os.listdir(src):
    if f.endswith("_kernel_stats.csv"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in os.listdir(src):
    if f.endswith("_counter_collection.csv"):
        for r in csv.DictReader(open(os.path.join(src, f))):
            name = r["Kernel_Name"]
            if name.startswith("void at::") or "rocclr" in name:
                continue
            pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in pmc.items():
    e = {}
    for c, v in d.items():
        big = [x for x in v if x >= 0.5 * max(v)]     # full-size launches only
        e[c + "_KiB_per_launch"] = sum(big) / len(big)
        e[c + "_launches"] = len(big)
    if "FETCH_SIZE_KiB_per_launch" in e and "WRITE_SIZE_KiB_per_launch" in e:
        e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KiB_per_launch"] + e["WRITE_SIZE_KiB_per_launch"]) * 1024
    out[k] = e
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
