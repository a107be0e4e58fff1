"""Quick throughput probe of the WAM workload (config 2 shapes): for every batch size on the command
line, `reps` independent batches iterated 100 times, serially (NSTREAMS=0) or on a stream pool.
Counts the iterations the runs actually made.   python scripts/quick_bench.py 1024,16384 [reps]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1024").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
if os.environ.get("FIELDS", "1") == "2":
    mod.SendCommand("computedistancefield kinbody mug")      # a second field (the reference demo has table and mug)
mod.set_num_streams(int(os.environ.get('NSTREAMS', '0')))
kw = dict(n_points=100, lambda_=100.0, obs_factor=float(os.environ.get('OBS_FACTOR', '500.0')))
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ORC_") or k == "NSTREAMS")
for n_runs in sizes:
    r = reps if n_runs <= 4096 else max(2, reps // 3)
    bids = [mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + k), **kw) for k in range(r + 1)]
    mod.batch_iterate(bids[0], 100)
    mod.kernel_time(reset=True)
    t0 = time.perf_counter()
    for b in bids[1:]:
        mod.batch_iterate_async(b, 100)
    for b in bids[1:]:
        mod.batch_sync(b)
    t1 = time.perf_counter()
    made = sum(int(mod.batch_iterations_done(b).sum()) for b in bids[1:])
    ms, n = mod.kernel_time()
    for b in bids:
        mod.batch_destroy(b)
    print("[%s] runs %d x %d launches: %.3f M it/s wall (iterations made), kernel avg %.2f ms" % (tag, n_runs, r, made / (t1 - t0) / 1e6, ms / n))
