"""Per-phase cycle counters of the iterate kernel for BASELINE config 4 or 5 (ORC_PHASE_TIMERS):
python scripts/phase_profile_cfg.py 4|5 [n_runs] [n_iter]"""
import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, time, ctypes as C
import common, or_cdchomp_amd
which = int(sys.argv[1]); n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
mod = or_cdchomp_amd.Module(0)
mod.set_workgroups_per_cu(int(os.environ.get('WGS_PER_CU', '0')))
mod.set_workgroup_threads(int(os.environ.get('WG_THREADS', '0')))
if which == 4:
    model = common.setup_product_wam(mod)
    goals, basegoals, seeds, kw = common.config4_problem(n_runs)
    bid = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
else:
    model = common.setup_product_tree30(mod)
    bid = mod.batch_create(model.name, common.config5_goals(n_runs), precision=int(os.environ.get("PRECISION", "32")), **dict(common.CONFIG5_KW, obs_factor=float(os.environ.get('OBS_FACTOR', '100.0'))))
mod.kernel_time(reset=True)
t0 = time.time(); costs, status = mod.batch_iterate(bid, n_iter); t1 = time.time()
ms, n = mod.kernel_time()
made = int(mod.batch_iterations_done(bid).sum())
out = np.zeros((n_runs, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "obs-reduce", "smooth+solve+step", "joint limits", "smooth cost"]
tot = out[:, :6].sum(1)
print("config %d: runs %d  wall %.1f ms kernel %.2f ms -> %.3g it/s (iterations made) ; mean cycles/iteration per WG %.0f ; status!=0: %d" % (
    which, n_runs, 1e3 * (t1 - t0), ms, made / (ms * 1e-3), tot.mean() / (n_iter + 1), int((status != 0).sum())))
for k in range(6):
    print("  %-18s %8.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean() / (n_iter + 1), 100 * out[:, k].sum() / tot.sum()))
q = np.percentile(tot, [0, 10, 50, 90, 99, 100]) / 1e6
print("per-WG total Mcycles: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(q))
rounds = out[:, 6]
print("limit rounds per run: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (rounds.mean(), np.median(rounds), np.percentile(rounds, 90), np.percentile(rounds, 99), rounds.max()))
pk = out[:, 7].astype(np.int64)
fast = pk & 0xFFFFF; scan = (pk >> 20) & 0xFFFFF; old = pk >> 40
print("joint-limit round kinds over all runs: closed form (<= 2 violated entries) %d, scans on register columns %d, general loop (>= 4 columns) %d" % (fast.sum(), scan.sum(), old.sum()))
