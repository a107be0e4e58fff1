"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines):
config 4 (floating base + WAM arm, n=14, n_points=200, momentum + hmc, batch 4096) and
config 5 (30-dof tree, 60 spheres, 4 fields at 1 cm cells, n_points=200, fp32, batch 4096).
Usage: python scripts/bench_configs.py [4|5] [n_runs] [n_iter]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
from or_cdchomp_amd import robots, scenes

which = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
mod = or_cdchomp_amd.Module(0)
if which == 4:
    model = common.setup_product_wam(mod)
    _, base, dofvals, adofs = common.wam_state()
    rng = np.random.default_rng(20250103)
    goals = common.wam_goals(n_runs, seed=20250103)
    basegoals = np.tile(np.asarray(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.3, 0.3, size=(n_runs, 3))
    kw = dict(n_points=200, lambda_=100.0, obs_factor=500.0, floating_base=1, use_momentum=1, use_hmc=1, hmc_resample_lambda=0.02)
    mk = lambda: mod.batch_create(model.name, goals, basegoals=basegoals, seeds=np.arange(n_runs, dtype=np.uint32), **kw)
    label = "config 4: floating base + arm, n=14, n_points=200, momentum+hmc, fp64"
else:
    model = robots.tree30()
    mod.add_robot(model, transform=[0, 0, 0, 0, 0, 0, 1.0], dof_values=np.zeros(model.n_dof), active_dofs=list(range(model.n_dof)))
    rng = np.random.default_rng(20250104)
    t0 = time.perf_counter()
    for name, (boxes, pose) in scenes.random_boxes(rng).items():
        mod.add_kinbody_boxes(name, boxes, transform=pose)
        mod.SendCommand("computedistancefield kinbody %s cube_extent 0.005 aabb_padding 0.15" % name)
    print("4 fields at 1 cm cells built in %.2f s" % (time.perf_counter() - t0))
    goals = np.random.default_rng(5).uniform(-0.8, 0.8, size=(n_runs, model.n_dof))
    kw = dict(n_points=200, lambda_=200.0, obs_factor=100.0)
    mk = lambda: mod.batch_create(model.name, goals, precision=32, **kw)
    label = "config 5: 30-dof tree, 60 spheres, 4 fields, n_points=200, fp32"
bids = [mk() for _ in range(3)]
mod.batch_iterate(bids[0], n_iter)
mod.kernel_time(reset=True)
t0 = time.perf_counter()
for b in bids[1:]:
    mod.batch_iterate_async(b, n_iter)
res = [mod.batch_sync(b) for b in bids[1:]]
t1 = time.perf_counter()
ms, n = mod.kernel_time()
bad = int(np.sum(res[0][1] != 0))
print("%s\n  batch %d x %d iterations: %.3f M it/s wall, kernel avg %.1f ms, runs with status != 0: %d" % (
    label, n_runs, n_iter, n_runs * n_iter * 2 / (t1 - t0) / 1e6, ms / n, bad))
