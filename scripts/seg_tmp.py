import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
import common, or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
n_runs = int(os.environ.get("NRUNS", "768"))
bid = mod.batch_create(model.name, common.wam_goals(n_runs), n_points=100, lambda_=100.0, obs_factor=500.0)
mod.batch_iterate(bid, 5)
mod.batch_iterate(bid, 100)
out = np.zeros((n_runs, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = sys.argv[1:]
for k in range(len(names)):
    print("%-40s %8.0f cycles per iteration" % (names[k], out[:, k].mean() / 100))
