"""it/s of the WAM holding the four-sphere box (config 2's goals, 1024 runs x 100 iterations): one launch at a time at both register
budgets, and eight launches on two streams at four workgroups per CU.   python scripts/held4_rate.py [n_runs=1024]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = []
for wgs, streams in ((0, 0), (4, 0), (4, 2), (0, 2)):
    mod = or_cdchomp_amd.Module(0)
    mod.set_num_streams(streams)
    mod.set_workgroups_per_cu(wgs)
    model, hand, pose = common.setup_product_wam_held4(mod)
    warm = mod.batch_create(model.name, common.wam_goals(n_runs, seed=5), **common.CONFIG2_KW)
    mod.batch_iterate(warm, 100); mod.batch_destroy(warm)
    n_b = 8 if streams else 4
    ids = [mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + k), **common.CONFIG2_KW) for k in range(n_b)]
    t0 = time.perf_counter()
    if streams:
        for b in ids: mod.batch_iterate_async(b, 100)
        for b in ids: mod.batch_sync(b)
    else:
        for b in ids: mod.batch_iterate(b, 100)
    t1 = time.perf_counter()
    made = sum(int(mod.batch_iterations_done(b).sum()) for b in ids)
    out.append("%s/CU %s: %.3g M it/s" % (wgs or 3, "two streams" if streams else "serial", made / (t1 - t0) / 1e6))
    mod.close()
print("held4 " + os.path.basename(os.environ.get("ORC_LIB", "product")) + ": " + "; ".join(out))
