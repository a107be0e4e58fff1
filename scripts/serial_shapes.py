import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, common, or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
for wgs, thr in ((0, 192), (4, 0), (0, 0)):
    mod.set_workgroups_per_cu(wgs); mod.set_workgroup_threads(thr)
    bids = [mod.batch_create(model.name, common.wam_goals(1024, seed=20250101 + k), **common.CONFIG2_KW) for k in range(11)]
    mod.batch_iterate(bids[0], 100)
    t0 = time.perf_counter()
    made = 0
    for b in bids[1:]:
        mod.batch_iterate(b, 100)
    t1 = time.perf_counter()
    made = sum(int(mod.batch_iterations_done(b).sum()) for b in bids[1:])
    for b in bids: mod.batch_destroy(b)
    print("wgs %d threads %d: %.3f M it/s one launch of 1024 at a time" % (wgs, thr, made / (t1 - t0) / 1e6))
