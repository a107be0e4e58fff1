"""Per-phase cycle counters (ORC_PHASE_TIMERS) of ONE run -- the reference's own use, SURVEY.md 8b: where the
25 us of an iteration go when a single workgroup has the chip to itself.   python scripts/phase_profile_single.py [n_iter]"""
import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, ctypes as C
import common, or_cdchomp_amd
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
goal = np.array([[0.6, -1.2, 0.3, 1.6, -0.4, 0.5, 0.2]])
bid = mod.batch_create(model.name, goal, n_points=100, lambda_=100.0, obs_factor=500.0)
mod.batch_iterate(bid, 5)
mod.kernel_time(reset=True)
mod.batch_iterate(bid, n_iter)
ms, n = mod.kernel_time()
out = np.zeros((1, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "obs-reduce", "smooth+solve+step", "joint limits", "smooth cost"]
tot = out[0, :6].sum()
print("one WAM run: kernel %.3f ms for %d iterations = %.2f us per iteration ; cycles per iteration (all calls since create: %d iterations) %.0f" % (ms, n_iter, 1e3 * ms / n_iter, n_iter + 5, tot / (n_iter + 5 + 2)))
for k in range(6):
    print("  %-18s %8.0f cycles/iter  %5.1f %%" % (names[k], out[0, k] / (n_iter + 5 + 2), 100 * out[0, k] / tot))
print("limit rounds: %d" % out[0, 6])
