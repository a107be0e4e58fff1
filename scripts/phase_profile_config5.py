import os, sys
os.environ["ORC_PHASE_TIMERS"] = "1"; os.environ["ORC_DEBUG_PLAN"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
import common, or_cdchomp_amd
from or_cdchomp_amd import robots, scenes
mod = or_cdchomp_amd.Module(0)
model = robots.tree30()
mod.add_robot(model, transform=[0, 0, 0, 0, 0, 0, 1.0], dof_values=np.zeros(model.n_dof), active_dofs=list(range(model.n_dof)))
rng = np.random.default_rng(20250104)
for name, (boxes, pose) in scenes.random_boxes(rng).items():
    mod.add_kinbody_boxes(name, boxes, transform=pose)
    mod.SendCommand("computedistancefield kinbody %s cube_extent 0.005 aabb_padding 0.15" % name)
n_runs = 512
goals = np.random.default_rng(5).uniform(-0.8, 0.8, size=(n_runs, model.n_dof))
bid = mod.batch_create(model.name, goals, precision=32, n_points=200, lambda_=200.0, obs_factor=100.0)
mod.batch_iterate(bid, 3)
mod.kernel_time(reset=True)
mod.batch_iterate(bid, 30)
ms, n = mod.kernel_time()
out = np.zeros((n_runs, 8))
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "obs-reduce", "smooth+solve+step", "joint limits", "smooth cost"]
tot = out[:, :6].sum(1)
print("kernel %.1f ms for %d runs x 30 iterations; mean cycles/iteration per WG %.0f" % (ms, n_runs, tot.mean()/31))
for k in range(6):
    print("  %-18s %9.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean()/31, 100*out[:, k].sum()/tot.sum()))
