"""waypoint-iterations/s across regimes (serial launches, 1024 runs x 50 iterations): looking for cliffs.  python scripts/regime_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
def rate(label, n_runs=1024, setup=common.setup_product_wam, goals=None, **kw):
    mod = or_cdchomp_amd.Module(0)
    model = setup(mod)
    if isinstance(model, tuple): model = model[0]
    g = common.wam_goals(n_runs, seed=20250101) if goals is None else goals
    warm = mod.batch_create(model.name, g, **kw); mod.batch_iterate(warm, 5); mod.batch_destroy(warm)
    bid = mod.batch_create(model.name, g, **kw)
    plan = mod.batch_plan(bid)
    t0 = time.perf_counter(); mod.batch_iterate(bid, 50); t1 = time.perf_counter()
    made = int(mod.batch_iterations_done(bid).sum())
    npts = kw.get("n_points", 101)
    print("%-44s %.3g M it/s  %.3g G waypoint-it/s  plan: variant %d, %d threads x %d/CU, tile %d of %d, solve %d" % (
        label, made / (t1 - t0) / 1e6, made * npts / (t1 - t0) / 1e9, plan["variant"], plan["threads"], plan["workgroups_per_cu"], plan["tile_m"], npts - 2, plan["solve_mode"]), flush=True)
    mod.batch_destroy(bid); mod.close()
base = dict(lambda_=100.0, obs_factor=500.0)
for npts in (8, 16, 30, 50, 66, 100, 130, 160):
    rate("WAM n_points %d" % npts, n_points=npts, **base)
rate("WAM 100, momentum", n_points=100, use_momentum=1, **base)
rate("WAM 100, momentum + hmc", n_points=100, use_momentum=1, use_hmc=1, hmc_resample_lambda=0.02, **base)
rate("WAM 100, derivative 2", n_points=100, derivative=2, **base)
rate("WAM 100, fp32", n_points=100, precision=32, **base)
rate("WAM 100, batch 4096", n_runs=4096, n_points=100, **base)
rate("WAM held4 100", setup=common.setup_product_wam_held4, n_points=100, **base)
