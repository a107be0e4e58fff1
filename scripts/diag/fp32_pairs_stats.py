"""fp32 in the pair-list family against fp32 in the many-sphere family (ORC_PAIRS_CHAIN64_ONLY=1), both against the product's own fp64:
how many of 512 runs (WAM holding the four-sphere box, 100 waypoints, 50 iterations) end further than 1e-4 / 1e-3 from fp64.
fp32 runs meet the reference's discontinuities (one-sided field interpolation, range tests, limit rounds) at a rounding of 6e-8."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, common, or_cdchomp_amd
n = 512
goals = common.wam_goals(n, seed=31)
goals[:, :7] = 0.6 * goals[:, :7] + 0.4 * np.asarray(common.wam_state()[2][:7])
kw = dict(n_points=100, lambda_=100.0, obs_factor=200.0)
if len(sys.argv) > 1: kw["obs_factor_self"] = float(sys.argv[1])
res = {}
for name, env, prec in (("pairs32", None, 32), ("generic32", "1", 32), ("pairs64", None, 64)):
    if env: os.environ["ORC_PAIRS_CHAIN64_ONLY"] = env
    else: os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None)
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    bid = mod.batch_create(model.name, goals, precision=prec, **kw) if prec == 32 else mod.batch_create(model.name, goals, **kw)
    c, s = mod.batch_iterate(bid, 50)
    res[name] = (mod.batch_gettraj(bid), c, s)
    mod.batch_destroy(bid); mod.close()
ok = (res["pairs64"][2] == 0)
for a in ("pairs32", "generic32"):
    okk = ok & (res[a][2] == 0)
    e = np.array([common.rel_l2(res[a][0][k], res["pairs64"][0][k]) for k in np.flatnonzero(okk)])
    print("%s vs fp64: %d runs, median %.1e, p90 %.1e, > 1e-5: %d, > 1e-4: %d, > 1e-3: %d, > 1e-2: %d ; status differs: %d" % (
        a, len(e), np.median(e), np.percentile(e, 90), (e > 1e-5).sum(), (e > 1e-4).sum(), (e > 1e-3).sum(), (e > 1e-2).sum(), (res[a][2] != res["pairs64"][2]).sum()))
