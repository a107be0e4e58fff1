"""it/s and per-phase cycles of long trajectories (WAM, n_points 300 / 400 / 600, 512 runs x 50 iterations, serial launches): derivative 1
beyond 256 moving waypoints solves the metric by parallel cyclic reduction (one barrier per level) where shorter runs take the closed-form
scans (one barrier).   python scripts/long_traj_rate.py"""
import os, sys, time
os.environ["ORC_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import ctypes as C
import numpy as np
import common, or_cdchomp_amd
n_runs = 512
for n_points in (200, 258, 300, 400, 600):
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    kw = dict(n_points=n_points, lambda_=100.0 * n_points / 100.0, obs_factor=500.0)
    warm = mod.batch_create(model.name, common.wam_goals(n_runs, seed=5), **kw); mod.batch_iterate(warm, 10); mod.batch_destroy(warm)
    bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101), **kw)
    plan = mod.batch_plan(bid)
    t0 = time.perf_counter(); mod.batch_iterate(bid, 50); t1 = time.perf_counter()
    made = int(mod.batch_iterations_done(bid).sum())
    ph = np.zeros((n_runs, 8))
    mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", ph.ctypes.data_as(C.POINTER(C.c_double)), ph.size))
    it = mod.batch_iterations_done(bid).astype(np.float64) + 1.0
    cyc = ph[:, :6].sum(axis=0) / it.sum()
    print("n_points %d: %.3g M it/s (%.3g M waypoint-iterations/s); plan solve_mode %d, %d threads, %d per CU, tile %d; cycles per iteration: FK %.0f cost %.0f update %.0f limits %.0f smooth cost %.0f; rounds per run-iteration %.2f"
          % (n_points, made / (t1 - t0) / 1e6, made * n_points / (t1 - t0) / 1e6, plan["solve_mode"], plan["threads"], plan["workgroups_per_cu"], plan["tile_m"],
             cyc[0], cyc[1] + cyc[2], cyc[3], cyc[4], cyc[5], ph[:, 6].sum() / it.sum()), flush=True)
    mod.batch_destroy(bid); mod.close()
