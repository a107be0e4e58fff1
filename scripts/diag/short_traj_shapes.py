"""short trajectories (8..50 waypoints) under the workgroup shapes the fp64 16-lane family is built for: is a smaller workgroup the better
plan when a run has few waypoints?   serial launches, 4096 runs x 50 iterations"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, common, or_cdchomp_amd
for npts in (8, 16, 24, 34, 50, 100):
    out = []
    for threads in (0, 128, 192):
        mod = or_cdchomp_amd.Module(0)
        mod.set_workgroup_threads(threads)
        model = common.setup_product_wam(mod)
        g = common.wam_goals(4096, seed=20250101)
        kw = dict(n_points=npts, lambda_=100.0, obs_factor=500.0)
        try:
            warm = mod.batch_create(model.name, g, **kw); mod.batch_iterate(warm, 5); mod.batch_destroy(warm)
            bid = mod.batch_create(model.name, g, **kw)
            plan = mod.batch_plan(bid)
            t0 = time.perf_counter(); mod.batch_iterate(bid, 50); t1 = time.perf_counter()
            made = int(mod.batch_iterations_done(bid).sum())
            out.append("%s: %.3g M it/s (%d thr x %d/CU, tile %d)" % (threads or "planner", made / (t1 - t0) / 1e6, plan["threads"], plan["workgroups_per_cu"], plan["tile_m"]))
        except RuntimeError as e:
            out.append("%s: %s" % (threads, str(e)[:40]))
        mod.close()
    print("n_points %d: " % npts + "; ".join(out), flush=True)
