import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, common, or_cdchomp_amd
from oracle import oracle_py as O
O.build(ref=False)
goals = common.wam_goals(12, seed=31)
goals[:, :7] = 0.6 * goals[:, :7] + 0.4 * np.asarray(common.wam_state()[2][:7])
kw = dict(n_points=100, lambda_=100.0, obs_factor=200.0)
res = {}
for name, env, prec in (("pairs32", None, 32), ("generic32", "1", 32), ("pairs64", None, 64)):
    if env: os.environ["ORC_PAIRS_CHAIN64_ONLY"] = env
    else: os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None)
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    bid = mod.batch_create(model.name, goals, precision=prec, **kw) if prec == 32 else mod.batch_create(model.name, goals, **kw)
    print(name, mod.batch_plan(bid))
    c, s = mod.batch_iterate(bid, 50)
    res[name] = (mod.batch_gettraj(bid), c, s)
    mod.batch_destroy(bid); mod.close()
for a, b in (("pairs32", "pairs64"), ("generic32", "pairs64"), ("pairs32", "generic32")):
    print(a, "vs", b, ["%.1e" % common.rel_l2(res[a][0][k], res[b][0][k]) for k in range(12)])
# per-iteration divergence of run 9: iterate step by step
for name, env in (("pairs32", None), ("generic32", "1")):
    if env: os.environ["ORC_PAIRS_CHAIN64_ONLY"] = env
    else: os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None)
    m32 = or_cdchomp_amd.Module(0); m64 = or_cdchomp_amd.Module(0)
    os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None)
    mo, _, _ = common.setup_product_wam_held4(m32); common.setup_product_wam_held4(m64)
    if env: os.environ["ORC_PAIRS_CHAIN64_ONLY"] = env
    b32 = m32.batch_create(mo.name, goals[9:10], precision=32, **kw); os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None); b64 = m64.batch_create(mo.name, goals[9:10], **kw)
    out = []
    for it in range(50):
        m32.batch_iterate(b32, 1); m64.batch_iterate(b64, 1)
        out.append(common.rel_l2(m32.batch_gettraj(b32)[0], m64.batch_gettraj(b64)[0]))
    print(name, "run 9 per iteration:", ["%.0e" % e for e in out])
    m32.close(); m64.close()
