"""Per-phase cycle counters (ORC_PHASE_TIMERS) of the TSR-constrained bench lines:  python scripts/phase_profile_tsr.py tsr1|tsr3 [n_runs] [n_iter]"""
import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, ctypes as C
import bench, or_cdchomp_amd
which = sys.argv[1]; n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024; n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
w = bench.Workload(which, 0, 1, n_runs)
mod = or_cdchomp_amd.Module(0)
mod.set_workgroups_per_cu(int(os.environ.get('WGS_PER_CU', '4')))
mod.set_workgroup_threads(int(os.environ.get('WG_THREADS', '0')))
w.setup(mod)
bid = w.create(mod, 0, 0)
mod.kernel_time(reset=True)
costs, status = mod.batch_iterate(bid, n_iter)
ms, n = mod.kernel_time()
out = np.zeros((n_runs, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "constraint step", "smooth+solve", "step + joint limits", "smooth cost"]
tot = out[:, :6].sum(1)
print("%s: runs %d kernel %.2f ms -> %.3g it/s ; mean cycles/iteration per WG %.0f ; status!=0: %d" % (which, n_runs, ms, n_runs * n_iter / (ms * 1e-3), tot.mean() / (n_iter + 1), int((status != 0).sum())))
for k in range(6):
    print("  %-20s %8.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean() / (n_iter + 1), 100 * out[:, k].sum() / tot.sum()))
