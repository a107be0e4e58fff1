"""config 2's batch with 1..4 fields in the scene (the extra ones far from the arm, every other one rotated): what a field costs the 16-lane pass"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, common, or_cdchomp_amd
from or_cdchomp_amd import scenes
for nf in (1, 2, 3, 4):
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    # (extra fields far away from the arm: the runs themselves do not change, only the lookups are made)
    for k in range(1, nf):
        rot = [0, 0, 0.38268, 0.92388] if (k % 2 == 0) else [0, 0, 0, 1]
        mod.add_kinbody_boxes("far%d" % k, [([0, 0, 0.3, 0, 0, 0, 1], [0.04, 0.04, 0.3])], transform=[6.0 + k, 5.0, 0.7] + rot)
        mod.SendCommand("computedistancefield kinbody far%d aabb_padding 0.15" % k)
    g = common.wam_goals(1024, seed=20250101)
    kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
    warm = mod.batch_create(model.name, g, **kw); mod.batch_iterate(warm, 5); mod.batch_destroy(warm)
    bid = mod.batch_create(model.name, g, **kw)
    plan = mod.batch_plan(bid)
    t0 = time.perf_counter(); mod.batch_iterate(bid, 50); t1 = time.perf_counter()
    made = int(mod.batch_iterations_done(bid).sum())
    print("%d field(s): %.3g M it/s, plan variant %d tile %d x%d/CU" % (nf, made/(t1-t0)/1e6, plan["variant"], plan["tile_m"], plan["workgroups_per_cu"]), flush=True)
    mod.close()
