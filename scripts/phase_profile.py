import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, time
import common
import or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
goals = common.wam_goals(n_runs)
kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
bid = mod.batch_create(model.name, goals, **kw)
mod.batch_iterate(bid, 5)
mod.kernel_time(reset=True)
t0 = time.time(); mod.batch_iterate(bid, 100); t1 = time.time()
ms, n = mod.kernel_time()
ph = mod._lib  # noqa
out = np.zeros((n_runs, 8))
import ctypes as C
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "obs-reduce", "smooth+solve+step", "joint limits", "smooth cost", "-", "-"]
tot = out[:, :6].sum(1)
print("runs %d  kernel %.2f ms  -> %.3g it/s ; mean cycles/iteration per WG %.0f" % (n_runs, ms, n_runs*100/(ms*1e-3), tot.mean()/101))
for k in range(6):
    print("  %-18s %8.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean()/101, 100*out[:, k].sum()/tot.sum()))
