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
bid = mod.batch_create(model.name, goals[1:2], **kw)
p = O.default_params(**kw)
run = O.OraRun(rob, base, dofvals, adofs, goals[1], [prob['sdf']], [prob['pose']], p)
import ctypes as C
L = O.lib()
for it in range(100):
    mod.batch_iterate(bid, 1)   # note: each call also runs a final eval, harmless
    traj = mod.batch_gettraj(bid)[0]
    tot=C.c_double(); ob=C.c_double(); sm=C.c_double()
    L.ora_chomp_iterate(L.ora_run_chomp(run.h), 1, C.byref(tot), C.byref(ob), C.byref(sm))
    e = common.rel_l2(traj, run.traj())
    lim = run.chomp().last_num_limadjs
    if lim or e > 1e-12 or it % 10 == 0:
        print(it, "%.2e" % e, "limadjs", lim, "T range", run.traj().min(0).round(3), run.traj().max(0).round(3))
