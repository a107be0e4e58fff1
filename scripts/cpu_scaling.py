import sys, time, os; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, common
from oracle import oracle_py as O
O.build(ref=False)
model, base, dofvals, adofs = common.wam_state()
prob = common.tabletop_problem(O); rob = O.OraRobot(model)
p = O.default_params(n_points=100, lambda_=100.0, obs_factor=500.0)
goals = common.wam_goals(256)
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup v2 cpu.max", e)
for t in (1, 4, 16, 64, 128):
    nr = min(256, max(8, 4*t))
    t0 = time.perf_counter(); _,_,_,thr = O.batch_run(rob, base, dofvals, adofs, goals[:nr], [prob["sdf"]], [prob["pose"]], p, 100, max_threads=t); t1 = time.perf_counter()
    print("threads %3d (used %d): %d runs in %.2f s -> %.0f it/s, %.0f it/s per thread" % (t, thr, nr, t1-t0, nr*100/(t1-t0), nr*100/(t1-t0)/thr))
