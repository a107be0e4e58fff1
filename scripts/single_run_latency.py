"""Wall time of the reference's own use -- one run through create / iterate / gettraj / destroy --
command by command.   python scripts/single_run_latency.py [n_iter]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
from or_cdchomp_amd import bindings
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
mod = bindings.bind(or_cdchomp_amd.Module(0))
model = common.setup_product_wam(mod)
goal = [0.6, -1.2, 0.3, 1.6, -0.4, 0.5, 0.2]
rows = []
for rep in range(12):
    t0 = time.perf_counter()
    run = mod.create(robot=model.name, adofgoal=goal, n_points=100, lambda_=100.0, obs_factor=500.0)
    t1 = time.perf_counter()
    mod.iterate(run=run, n_iter=n_iter)
    t2 = time.perf_counter()
    text = mod.gettraj(run=run, no_collision_check=True)
    t3 = time.perf_counter()
    text = mod.gettraj(run=run, no_collision_exception=True)
    t4 = time.perf_counter()
    mod.destroy(run=run)
    t5 = time.perf_counter()
    rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4])
r = np.median(np.array(rows[2:]), axis=0) * 1e3
print("one WAM run, 100 waypoints, %d iterations (median of 10, ms): create %.3f  iterate %.3f (%.1f us per iteration)  gettraj %.3f  "
      "gettraj with the collision re-check %.3f  destroy %.3f  ; total %.3f" % (n_iter, r[0], r[1], 1e3 * r[1] / max(n_iter, 1), r[2], r[3], r[4], r[0] + r[1] + r[3] + r[4]))
