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
goals = common.wam_goals(3, seed=7)
model_, base, dofvals, adofs = common.wam_state()
for kw in [dict(obs_factor=0.0, obs_factor_self=0.0), dict(obs_factor=500.0, obs_factor_self=0.0), dict(obs_factor=0.0, obs_factor_self=10.0), dict(obs_factor=500.0, obs_factor_self=10.0)]:
    kw.update(n_points=100, lambda_=100.0)
    bid = mod.batch_create(model.name, goals, **kw)
    mod.batch_iterate(bid, 1)
    Gd = mod.batch_state(bid, "G"); AGd = mod.batch_state(bid, "AG"); trd = mod.batch_trace(bid, 1)
    p = O.default_params(**kw)
    for k in range(1):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob['sdf']], [prob['pose']], p)
        st, costs, tr = run.iterate(1, trace=True)
        G = run.mat("G", run.m, run.n).copy() * run.m; AG = run.mat("AG", run.m, run.n).copy()
        print(kw, "G rel", common.rel_l2(Gd[k], G), "AG rel", common.rel_l2(AGd[k], AG), "trace", trd[k,0], tr[0])
        err = np.abs(Gd[k]-G); i = np.unravel_index(err.argmax(), err.shape); print("  worst at", i, Gd[k][i], G[i], "row err", np.linalg.norm(Gd[k]-G, axis=1).round(6)[:12])
    mod.batch_destroy(bid)
