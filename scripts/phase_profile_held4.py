"""Per-phase cycle counters (ORC_PHASE_TIMERS) and rate of the WAM holding the four-sphere box, config 2's goals:
python scripts/phase_profile_held4.py [n_runs=1024] [n_iter=100]"""
import sys, os, time
os.environ["ORC_PHASE_TIMERS"] = "1"
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, ctypes as C
import common, or_cdchomp_amd
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mod = or_cdchomp_amd.Module(0)
mod.set_workgroups_per_cu(int(os.environ.get('WGS_PER_CU', '0')))
mod.set_workgroup_threads(int(os.environ.get('WG_THREADS', '0')))
model, hand, pose = common.setup_product_wam_held4(mod)
for step in range(3):
    bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + step), **common.CONFIG2_KW)
    mod.kernel_time(reset=True)
    costs, status = mod.batch_iterate(bid, n_iter)
    ms, n = mod.kernel_time()
    made = int(mod.batch_iterations_done(bid).sum())
    out = np.zeros((n_runs, 8))
    mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
    mod.batch_destroy(bid)
names = ["FK", "cost", "constraint step", "smooth+solve+step", "joint limits", "smooth cost"]
tot = out[:, :6].sum(1)
print("held4: runs %d kernel %.2f ms -> %.3g it/s (iterations made) ; mean cycles/iteration per WG %.0f ; status!=0: %d" % (
    n_runs, ms, made / (ms * 1e-3), tot.mean() / (n_iter + 1), int((status != 0).sum())))
for k in range(6):
    print("  %-18s %8.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean() / (n_iter + 1), 100 * out[:, k].sum() / tot.sum()))
q = np.percentile(tot, [0, 10, 50, 90, 99, 100]) / 1e6
print("per-WG total Mcycles: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(q))
