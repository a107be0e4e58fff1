import os, sys
os.environ["ORC_HMC_DEVICE"] = "1"; os.environ["ORC_DEBUG_VERDICT"] = "1"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, or_cdchomp_amd
for dev in (0, [0, 0, 0]):
    mod = or_cdchomp_amd.Module(dev)
    model = common.setup_product_wam(mod)
    goals, basegoals, seeds, kw = common.config4_problem(37)
    kw = dict(kw, n_points=50, hmc_resample_lambda=0.2)
    bid = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
    c, s = mod.batch_iterate(bid, 20)
    sys.stderr.write("module %r\n" % (dev,)); sys.stderr.flush()
    v = mod.batch_collision_verdict(bid)
    print(dev, v["collides"].sum())
