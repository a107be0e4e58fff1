"""it/s of the WAM with its three finger dofs active (a tree) holding the four-sphere box, and of the same runs in fp32: the pair-list
family (round 6) against the many-sphere family they took until round 5 (ORC_PAIRS_CHAIN64_ONLY=1).  1024 runs x 100 iterations, two streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
n_runs = 1024
def goals(seed, fingers):
    arm = common.wam_goals(n_runs, seed=seed)
    return np.ascontiguousarray(np.hstack([arm, np.random.default_rng(seed + 7).uniform(0.2, 2.2, size=(n_runs, 3))])) if fingers else arm
for label, fingers, prec in (("tree fp64", 1, 64), ("chain fp32", 0, 32), ("tree fp32", 1, 32)):
    out = []
    for fam, env in (("pair list", None), ("many-sphere", "1")):
        if env: os.environ["ORC_PAIRS_CHAIN64_ONLY"] = env
        else: os.environ.pop("ORC_PAIRS_CHAIN64_ONLY", None)
        for streams in (0, 2):
            mod = or_cdchomp_amd.Module(0)
            mod.set_num_streams(streams)
            model, hand, pose = common.setup_product_wam_held4(mod)
            if fingers: mod.set_active_dofs(model.name, list(range(10)))
            kw = dict(common.CONFIG2_KW)
            if prec == 32: kw["precision"] = 32
            warm = mod.batch_create(model.name, goals(5, fingers), **kw); mod.batch_iterate(warm, 100); mod.batch_destroy(warm)
            n_b = 8 if streams else 4
            ids = [mod.batch_create(model.name, goals(20250101 + k, fingers), **kw) for k in range(n_b)]
            t0 = time.perf_counter()
            if streams:
                for b in ids: mod.batch_iterate_async(b, 100)
                for b in ids: mod.batch_sync(b)
            else:
                for b in ids: mod.batch_iterate(b, 100)
            t1 = time.perf_counter()
            made = sum(int(mod.batch_iterations_done(b).sum()) for b in ids)
            out.append("%s %s %.3g M" % (fam, "two streams" if streams else "serial", made / (t1 - t0) / 1e6))
            mod.close()
    print("held4 %s: " % label + "; ".join(out), flush=True)
