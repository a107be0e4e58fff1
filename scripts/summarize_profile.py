#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by scripts/profile_r01.sh) into the
tracked summaries under profiles/: the kernel-trace stats CSV, a PMC summary JSON, and
profiles/traffic_latest.json (HBM bytes per iterate launch, read by bench.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]          # e.g.  a  r01_baseline
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
summary = {"source": "rocprofv3 on `python3 bench.py --steps N --warmup 1 --no-cpu-baseline` (scripts/profile_r01.sh)",
           "kernel": "chomp_iterate_kernel<double>", "counters": {}}
for row in csv.DictReader(open(stats)):
    if "chomp_iterate" in row["Name"]:
        summary["kernel_trace"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                   "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(files[0])):
        if "chomp_iterate" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"VGPR_Count": r["VGPR_Count"], "Accum_VGPR_Count": r["Accum_VGPR_Count"],
                    "SGPR_Count": r["SGPR_Count"], "LDS_Block_Size": r["LDS_Block_Size"],
                    "Scratch_Size": r["Scratch_Size"], "Grid_Size": r["Grid_Size"], "Workgroup_Size": r["Workgroup_Size"]}
    for k, v in agg.items():
        summary["counters"][k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    summary["dispatch"] = meta
c = summary["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
    # of a wide coalesced read -> doubled (our gathers are 8-byte, so this is an upper estimate)
    fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    write = c["WRITE_SIZE"]["mean_per_launch"] * 1024
    summary["hbm_bytes_per_launch"] = fetch + write
    json.dump({"batch": 1024, "n_iter": 100, "hbm_bytes_per_launch": fetch + write,
               "fetch_bytes_corrected": fetch, "write_bytes": write, "from": name},
              open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
json.dump(summary, open(os.path.join(dst, name + "_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
