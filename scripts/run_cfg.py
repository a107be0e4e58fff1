"""One iterate call of a BASELINE configuration (2, 4 or 5) for profiling: python scripts/run_cfg.py <config> [n_runs] [n_iter]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
which = int(sys.argv[1]); n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else {2: 1024}.get(which, 4096); n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 100
mod = or_cdchomp_amd.Module(0)
mod.set_workgroups_per_cu(int(os.environ.get('WGS_PER_CU', '0')))
mod.set_workgroup_threads(int(os.environ.get('WG_THREADS', '0')))
if which == 4:
    model = common.setup_product_wam(mod)
    goals, basegoals, seeds, kw = common.config4_problem(n_runs)
    mk = lambda: mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
elif which == 5:
    model = common.setup_product_tree30(mod)
    mk = lambda: mod.batch_create(model.name, common.config5_goals(n_runs), precision=32, **common.CONFIG5_KW)
else:
    model = common.setup_product_wam(mod)
    mk = lambda: mod.batch_create(model.name, common.wam_goals(n_runs), **common.CONFIG2_KW)
bid = mk()
mod.kernel_time(reset=True)
t0 = time.time(); costs, status = mod.batch_iterate(bid, n_iter); t1 = time.time()
ms, n = mod.kernel_time()
made = int(mod.batch_iterations_done(bid).sum())
print("config %d: runs %d x %d iterations: kernel %.2f ms -> %.4g it/s (iterations made %d); status != 0: %d" % (
    which, n_runs, n_iter, ms, made / (ms * 1e-3), made, int((status != 0).sum())))
