"""config 4 with ORC_PHASE_TIMERS: per-run joint-limit rounds, per-phase cycles, status -> gpurun_out/r05/c4_rounds.npz"""
import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np, ctypes as C
import common, or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
n_runs = 4096
goals, basegoals, seeds, kw = common.config4_problem(n_runs)
bid = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
costs, status = mod.batch_iterate(bid, 100)
out = np.zeros((n_runs, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
trace = mod.batch_trace(bid, 100)
np.savez(os.path.join(os.path.dirname(__file__), "..", "..", "gpurun_out", "r05", "c4_rounds.npz"), phase=out, status=status, goals=goals,
         iters=mod.batch_iterations_done(bid), lo=np.asarray(model.limit_lower[:7]), hi=np.asarray(model.limit_upper[:7]))
print("dumped", out[:, 6].mean())
