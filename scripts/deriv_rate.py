"""it/s of config 2's WAM batch (1024 runs x 100 iterations) with `derivative` 1, 2 and 3 (src/libcd/chomp.c:239-340: the smoothness metric
of a higher derivative is penta- / hepta-diagonal), serial launches and two streams.   python scripts/deriv_rate.py [n_runs=1024] [lambda=100]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
for D in (1, 2, 3):
    out = []
    for streams in (0, 2):
        mod = or_cdchomp_amd.Module(0)
        mod.set_num_streams(streams)
        model = common.setup_product_wam(mod)
        kw = dict(common.CONFIG2_KW, derivative=D, lambda_=lam)
        warm = mod.batch_create(model.name, common.wam_goals(n_runs, seed=5), **kw)
        mod.batch_iterate(warm, 100); mod.batch_destroy(warm)
        n_b = 8 if streams else 4
        ids = [mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + k), **kw) for k in range(n_b)]
        t0 = time.perf_counter()
        if streams:
            for b in ids: mod.batch_iterate_async(b, 100)
            for b in ids: mod.batch_sync(b)
        else:
            for b in ids: mod.batch_iterate(b, 100)
        t1 = time.perf_counter()
        made = sum(int(mod.batch_iterations_done(b).sum()) for b in ids)
        bad = sum(int((mod.batch_iterate(b, 0)[1] != 0).sum()) for b in ids)
        out.append("%s: %.3g M it/s (%d of %d runs stopped)" % ("two streams" if streams else "serial", made / (t1 - t0) / 1e6, bad, n_b * n_runs))
        mod.close()
    print("derivative %d, lambda %g, %s: " % (D, lam, os.path.basename(os.environ.get("ORC_LIB", "product"))) + "; ".join(out), flush=True)
