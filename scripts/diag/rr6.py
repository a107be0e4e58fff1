import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, or_cdchomp_amd
from oracle import oracle_py as O
import test_gpu_random_robots as T
O.build(ref=False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 6
# replay the draw
rng = np.random.default_rng(5000 + seed)
model, what = T.random_robot(seed)
n_dof = model.n_dof
if rng.uniform() < 0.35 and n_dof > 3:
    adofs = sorted(rng.choice(n_dof, size=int(rng.integers(2, n_dof)), replace=False).tolist())
else:
    adofs = list(range(n_dof))
lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)]); hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
dofvals = rng.uniform(0.5 * lo, 0.5 * hi)
which = ("table", 2, 4, "table")[int(rng.integers(0, 4))]
base = ([-0.55, 0.05, 0.75] if which == "table" else [0.05, -0.1, 0.35]) + list(T._random_quat(rng, 0.7))
floating = bool(rng.uniform() < 0.25); precision = 32 if rng.uniform() < 0.25 else 64
momentum = bool(rng.uniform() < 0.3); second_order = bool(rng.uniform() < 0.12); hmc = momentum and bool(rng.uniform() < 0.4)
long_traj = bool(rng.uniform() < 0.1)
n_runs = (3, 3, 3, 40, 300)[int(rng.integers(0, 5))]
n_points = int(rng.integers(100, 230)) if long_traj else int(rng.integers(5, 72))
n_iter = int(rng.integers(6, 16))
kw = dict(n_points=n_points, lambda_=float(rng.uniform(120.0, 400.0)), obs_factor=float(rng.uniform(20.0, 200.0)),
          obs_factor_self=float(rng.uniform(2.0, 20.0)), epsilon=float(rng.uniform(0.06, 0.14)), epsilon_self=float(rng.uniform(0.02, 0.08)))
print(what, which, floating, precision, momentum, second_order, hmc, n_runs, n_points, n_iter, kw)
seeds = rng.integers(0, 1000, size=n_runs).astype(np.uint32)
shards = int(rng.integers(2, 4)) if rng.uniform() < 0.15 else 1
mod = or_cdchomp_amd.Module(0)
mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
grids, poses = T._scene(mod, O, which)
threads = (0, 0, 192, 512)[int(rng.integers(0, 4))]
per_cu = 4 if rng.uniform() < 0.3 and threads == 0 else 0
if rng.uniform() < 0.2: pass
goals = rng.uniform(0.7 * lo[adofs], 0.7 * hi[adofs], size=(n_runs, len(adofs)))
rob = O.OraRobot(model)
for thr in (threads, 0, 192):
    mod.set_workgroup_threads(thr)
    for it in (1, 3, n_iter):
        bid = mod.batch_create(model.name, goals, **kw)
        costs, status = mod.batch_iterate(bid, it)
        traj = mod.batch_gettraj(bid); mod.batch_destroy(bid)
        ot, oc, ost, _ = O.batch_run(rob, base, dofvals, adofs, goals, grids, poses, O.default_params(**kw), it)
        pt, _, _, _ = O.batch_run(rob, base, dofvals, adofs, goals * (1 + 2.0**-52), grids, poses, O.default_params(**kw), it)
        print("threads", thr, "iters", it, "err", ["%.2e" % common.rel_l2(traj[k], ot[k]) for k in range(n_runs)], "amp", ["%.2e" % common.rel_l2(pt[k], ot[k]) for k in range(n_runs)], status.tolist(), ost.tolist())
