#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (written by scripts/profile_round.sh) into the
tracked summaries under profiles/: the kernel-trace stats CSV, a PMC summary JSON, and the entry of
profiles/counters_latest.json that bench.py reads (HBM bytes per iterate launch, vector
wave-instructions per run-iteration).

    python scripts/summarize_profile.py <tag> <name> [config=2] [batch] [n_iter=100]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, name = sys.argv[1], sys.argv[2]          # e.g.  a  r02_config2
config = (int(sys.argv[3]) if sys.argv[3].isdigit() else sys.argv[3]) if len(sys.argv) > 3 else 2      # 2, 4, 5, tsr1, tsr3
batch = int(sys.argv[4]) if len(sys.argv) > 4 else {2: 1024, 4: 4096, 5: 4096, "tsr1": 1024, "tsr3": 1024, "held4": 1024}[config]
n_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 100
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
# gpurun merges every call's files into the same directory: the newest file is this call's
newest = lambda pattern: max(glob.glob(pattern), key=os.path.getmtime)
stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, name + "_kernel_stats.csv"))
summary = {"source": "rocprofv3 on `python3 bench.py --config %s --no-cpu-baseline --serial-steps 0 --steps N --warmup W` "
                     "(scripts/profile_round.sh): --kernel-trace --stats with 20 steps; one --pmc pass per counter group with 2 steps" % config,
           "config": config, "batch": batch, "n_iter": n_iter, "counters": {}}
for row in csv.DictReader(open(stats)):
    if "chomp_iterate" in row["Name"]:
        summary["kernel"] = row["Name"][:120]
        summary["kernel_trace"] = {"calls": int(row["Calls"]), "avg_ns": float(row["AverageNs"]),
                                   "min_ns": float(row["MinNs"]), "max_ns": float(row["MaxNs"])}
log = os.path.join(src, "bench_trace.log")
if os.path.exists(log):
    for line in open(log):
        if line.startswith("{"):
            summary["bench_line_of_the_traced_run"] = json.loads(line)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    files = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))
    if not files:
        continue
    files = [max(files, key=os.path.getmtime)]
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
sys.path.insert(0, root)
from or_cdchomp_amd import _capi
# the build the counters belong to: bench.py quotes them only while csrc/ still hashes to this
hfile = os.path.join(src, "csrc_hash.txt")      # written on the GPU box by scripts/profile_round.sh beside the counters
build_hash = open(hfile).read().strip() if os.path.exists(hfile) else _capi.csrc_hash()
entry = {"batch": batch, "n_iter": n_iter, "from": name, "source": "profiles/%s_summary.json" % name, "csrc_hash": build_hash}
summary["csrc_hash"] = entry["csrc_hash"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md "HBM": FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
    # of a wide coalesced read -> doubled (our gathers are narrow, so this is an upper estimate)
    fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
    write = c["WRITE_SIZE"]["mean_per_launch"] * 1024
    summary["hbm_bytes_per_launch"] = fetch + write
    entry.update(hbm_bytes_per_launch=fetch + write, fetch_bytes_corrected=fetch, write_bytes=write)
    # what the bytes are: the results of a launch (trajectories, momentum, costs, trace) are a few MB; everything a launch WRITES
    # beyond them is register state going to scratch around the phase calls (callee-saved registers, spills), and most of what it
    # reads is the same bytes coming back
    results = {2: 8.0 * 1024 * (100 * 7 + 98 * 7 + 3 * 100 + 8), "held4": 8.0 * 1024 * (100 * 7 + 98 * 7 + 3 * 100 + 8)}.get(config)
    entry["traffic_note"] = ("of the %.2f GB per launch %.2f GB are writes against ~%s of results: scratch traffic of the phase calls' callee-saved "
                             "registers and spills (and its way back), not the run's state, which stays in LDS; the field sits in L2"
                             % ((fetch + write) / 1e9, write / 1e9, ("%.0f MB" % (results / 1e6)) if results else "a few MB to tens of MB"))
    summary["traffic_note"] = entry["traffic_note"]
if "SQ_INSTS_VALU" in c:
    # the final cost-only pass of a launch is charged to its n_iter iterations
    entry["valu_insts_per_run_iteration"] = c["SQ_INSTS_VALU"]["mean_per_launch"] / (batch * n_iter)
    summary["valu_insts_per_run_iteration"] = entry["valu_insts_per_run_iteration"]
path = os.path.join(dst, "counters_latest.json")
allc = json.load(open(path)) if os.path.exists(path) else {}
allc["config%s" % config] = entry
json.dump(allc, open(path, "w"), indent=1)
json.dump(summary, open(os.path.join(dst, name + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "bench_line_of_the_traced_run"}, indent=1))
