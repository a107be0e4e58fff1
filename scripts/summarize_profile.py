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
    # what the bytes are.  The results of a launch (trajectories, momentum, costs, trace) are a few MB.  The plan of the traced run
    # (ORC_DEBUG_PLAN line in the log) says where the trajectory T and the cost pass's gradient rows G live: in global memory they
    # make round trips every iteration (FK reads T, the update phase stages it in and writes it back; the cost pass writes G, the
    # update phase reads it) -- through L2, of which the counters see what misses and what is written back.  What a launch writes
    # beyond results and those rows is register state going to scratch around the phase calls (callee-saved registers, spills).
    plan = None
    if os.path.exists(log):
        for line in open(log, errors="replace"):
            if line.startswith("orc plan:"):
                plan = line.strip()
    dims = {2: (100, 7, 8), "held4": (100, 7, 8), "tsr1": (100, 7, 8), "tsr3": (100, 7, 8), 4: (200, 14, 8), 5: (200, 30, 4)}.get(config)
    state = None
    if plan and dims:
        np_, n_, w_ = dims
        t_glob = " t_in_lds 0" in plan; g_glob = " g_in_lds 0" in plan
        per_it = (3 * np_ * n_ * w_ if t_glob else 0) + (2 * (np_ - 2) * n_ * w_ if g_glob else 0)
        state = per_it * batch * n_iter
    entry["traffic_note"] = ("%.2f GB per launch (%.2f GB of them writes) against a few MB of results.  Plan of the traced run: %s.  "
                             "By the layout's arithmetic %s; the rest of the writes is register state going to scratch around the phase calls "
                             "(callee-saved registers, spills) and%s its way back: not algorithmic traffic"
                             % ((fetch + write) / 1e9, write / 1e9, plan or "not in the log",
                                ("%.2f GB per launch are the trajectory's and gradient rows' round trips through global memory (3 x T + 2 x G per iteration where the plan keeps them out of LDS)"
                                 % (state / 1e9)) if state else "the run's state stays in LDS for the launch",
                                " (for the TSR lines) the constraint step's per-run workspace in global memory, and" if str(config).startswith("tsr") else ""))
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
