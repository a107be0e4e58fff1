import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common
from oracle import oracle_py as O
import or_cdchomp_amd
O.build(ref=False)
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
prob = common.tabletop_problem(O)
rob = O.OraRobot(model)
goals = common.wam_goals(8)
model_, base, dofvals, adofs = common.wam_state()
kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0, use_momentum=1)
for n_iter in [1, 2, 5, 10, 20, 50, 100]:
    bid = mod.batch_create(model.name, goals, **kw)
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    p = O.default_params(**kw)
    errs = []
    for k in range(8):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob['sdf']], [prob['pose']], p)
        st, c = run.iterate(n_iter)
        errs.append(common.rel_l2(traj[k], run.traj()))
    print(n_iter, ["%.1e" % e for e in errs], costs[0], c)
