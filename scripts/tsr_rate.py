"""Throughput of the TSR-constrained iteration (csrc/tsr.h): WAM, n_points=100, three constrained rows on every
moving point (a 294 x 294 system per run and iteration in the reference's dense form; ORC_TSR_DENSE=1 runs that form).   python scripts/tsr_rate.py [n_runs] [k rows 1..3]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
from or_cdchomp_amd import robots
from oracle import oracle_py as O
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 3
O.build(ref=False)
s2 = np.sqrt(0.5)
base = [-1.0, 0.0, 1.0, 0.0, s2, 0.0, s2]
mod = or_cdchomp_amd.Module(0)
model, _, dofvals, adofs = common.wam_state()
mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
from or_cdchomp_amd import scenes
scenes.add_tabletop(mod)
mod.SendCommand("computedistancefield kinbody table")
R, t, _, _ = O.OraRobot(model).fk(base, dofvals)
li = model.link_names.index("wam7")
Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0] if rows > 1 else [-3, 3], [0, 0] if rows > 2 else [-3, 3], [-3, 3]]
tsr = robots.Tsr(T0w_R=R[li], T0w_d=t[li], Bw=Bw)
rng = np.random.default_rng(1)
goals = np.array(robots.WAM_START)[None, :] + 0.4 * rng.uniform(-1, 1, size=(n_runs, 7))
bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points 100 lambda 100 obs_factor 200 con_tsr 'all link wam7' '%s'"
                          % (model.name, n_runs, goals.ctypes.data, tsr.serialize())))
mod.batch_iterate(bid, 2)
t0 = time.perf_counter(); costs, status = mod.batch_iterate(bid, 20); t1 = time.perf_counter()
print("TSR: %d runs x 20 iterations, %d constrained rows per point (system %d x %d): %.1f ms -> %.3g it/s ; status!=0: %d"
      % (n_runs, rows, 98 * rows, 98 * rows, (t1 - t0) * 1e3, n_runs * 20 / (t1 - t0), int((status != 0).sum())))
