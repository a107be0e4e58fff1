import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
mod.set_num_streams(int(os.environ.get('NSTREAMS', '0')))
kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
bids = [mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + k), **kw) for k in range(reps + 1)]
mod.batch_iterate(bids[0], 100)
mod.kernel_time(reset=True)
t0 = time.perf_counter()
for b in bids[1:]:
    mod.batch_iterate_async(b, 100)
for b in bids[1:]:
    mod.batch_sync(b)
t1 = time.perf_counter()
ms, n = mod.kernel_time()
print("runs %d: %.3f M it/s wall, kernel avg %.2f ms" % (n_runs, n_runs * 100 * reps / (t1 - t0) / 1e6, ms / n))
