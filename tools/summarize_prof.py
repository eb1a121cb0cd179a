#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_rNN/<run>/<host>/<pid>_*.csv) into the
tracked profiles/ directory:

    python tools/summarize_prof.py gpurun_out/prof_r02 r02

  profiles/rNN_<run>_kernel_stats.csv   the `--kernel-trace --stats` summaries (eval, train, bf16, tpsf)
  profiles/rNN_pmc_summary.json         per kernel: FETCH_SIZE / WRITE_SIZE per launch and HBM bytes per launch
  profiles/rNN_sq_summary.json          per kernel: SQ counters per launch and the derived shares
  profiles/rNN_clock_summary.json       per kernel: effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration
                                        (MI355X_MICROARCH.md 'DVFS give-back'; launches of >= 0.3 ms only)

HBM bytes per launch follow MI355X_MICROARCH.md section HBM: the counters are in KiB, and on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide (16 B/lane) coalesced reads, so traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
Full-size launches only (>= half the largest value of a kernel: the warm-up / B=32 launches are dropped).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
only = set(sys.argv[3].split()) if len(sys.argv) > 3 else None      # e.g. "tpsf": re-summarise that group only (its raw
#                                   data was re-collected on newer sources; the other groups keep their recorded hashes)
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
os.makedirs(dst, exist_ok=True)


def files(run, suffix):
    """Files of the NEWEST process of a run only: gpurun merges gpurun_out/ into the local copy, so an earlier
    collection's <pid>_*.csv files are still lying next to the new ones."""
    fs = glob.glob(os.path.join(src, run, "**", "*" + suffix), recursive=True)
    if not fs:
        return fs
    newest = max(fs, key=os.path.getmtime)
    pid = os.path.basename(newest).split("_")[0]
    return [f for f in fs if os.path.basename(f).split("_")[0] == pid]


for run in sorted(os.listdir(src)):
    if only is not None and run.split("_")[0] not in only:
        continue
    if run.endswith("_stats") and os.path.isdir(os.path.join(src, run)):
        for f in files(run, "_kernel_stats.csv"):
            shutil.copy(f, os.path.join(dst, f"{tag}_{run[:-6]}_kernel_stats.csv"))


meta = {}
if os.path.exists(os.path.join(src, "meta.json")):
    meta = json.load(open(os.path.join(src, "meta.json")))
try:
    import subprocess
    meta["summarized_at_commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                                                  cwd=os.path.dirname(dst)).stdout.strip()
except OSError:
    pass


def collect(runs):
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for run in runs:
        for f in files(run, "_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if name.startswith("void at::") or "rocclr" in name:
                    continue
                pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return pmc


def per_launch(v):
    big = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
    return sum(big) / len(big), len(big)


for mode in ("eval", "train", "tpsf", "bf16", "trainbf16"):
    if only is not None and mode not in only:
        continue
    out = {}
    for k, d in collect([f"{mode}_fetch", f"{mode}_write"]).items():
        e = {}
        for c, v in d.items():
            e[c + "_KiB_per_launch"], e[c + "_launches"] = per_launch(v)
        if "FETCH_SIZE_KiB_per_launch" in e and "WRITE_SIZE_KiB_per_launch" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE_KiB_per_launch"] + e["WRITE_SIZE_KiB_per_launch"]) * 1024
        out[k] = e
    if out:
        out["_meta"] = dict(meta)         # kernel-source hash of the build the counters were collected on
        name = f"{tag}_pmc_summary.json" if mode == "eval" else f"{tag}_{mode}_pmc_summary.json"
        json.dump(out, open(os.path.join(dst, name), "w"), indent=1, sort_keys=True)

sq = {}
for mode in ("eval", "train", "bf16"):
    if only is not None and mode not in only:
        continue
    for k, d in collect([f"{mode}_sq"]).items():
        e = {c: per_launch(v)[0] for c, v in d.items()}
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles
            e["wait_any_share"] = e.get("SQ_WAIT_ANY", 0) / wc
            e["wait_inst_share"] = e.get("SQ_WAIT_INST_ANY", 0) / wc
            e["active_inst_share"] = e.get("SQ_ACTIVE_INST_ANY", 0) / wc
            e["lds_bank_conflict_share_of_lds_cycles"] = (e.get("SQ_LDS_BANK_CONFLICT", 0) / e["SQ_LDS_IDX_ACTIVE"]
                                                         if e.get("SQ_LDS_IDX_ACTIVE") else None)
        sq[f"{mode}: {k}"] = e
if sq:      # a group re-collected on newer sources gets files of its own (rNN_<group>_sq_summary.json)
    name = f"{tag}_sq_summary.json" if only is None else f"{tag}_{'_'.join(sorted(only))}_sq_summary.json"
    json.dump(sq, open(os.path.join(dst, name), "w"), indent=1, sort_keys=True)
print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(tag)))

clk = {}
for mode in ("eval", "train", "bf16"):
    if only is not None and mode not in only:
        continue
    per = collections.defaultdict(list)
    for f in files(f"{mode}_clk", "_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Kernel_Name"].startswith("void at::"):
                continue
            dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if dur >= 300000:
                per[r["Kernel_Name"]].append((float(r["Counter_Value"]) / 8 / dur, dur))
    for k, v in per.items():
        clk[f"{mode}: {k}"] = {"effective_clock_GHz": round(sum(x for x, _ in v) / len(v), 3),
                               "avg_launch_ms": round(sum(d for _, d in v) / len(v) / 1e6, 3), "launches": len(v)}
if clk:
    name = f"{tag}_clock_summary.json" if only is None else f"{tag}_{'_'.join(sorted(only))}_clock_summary.json"
    json.dump(clk, open(os.path.join(dst, name), "w"), indent=1, sort_keys=True)
    print("wrote", name)
