"""CHOMP iterations/s over the batch sizes SURVEY.md 8(d) lists (WAM config, 100 iterations per
launch, one launch per batch size after a warm-up launch).  Writes one JSON object to stdout."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
out = {"workload": "WAM 7-DOF, n_points=100, n_iter=100, lambda=100, obs_factor=500, tabletop SDF", "dtype": "f64", "points": []}
for shape in (0, 192, 512):      # every kernel variant once before anything is timed
    mod.set_workgroup_threads(shape)
    warm = mod.batch_create(model.name, common.wam_goals(64, seed=1), **kw); mod.batch_iterate(warm, 10); mod.batch_destroy(warm)
# (batch, workgroup threads): the module-wide shape a caller would set for that batch size
# (orc_set_workgroup_threads: 512 = one run per CU on eight wavefronts for batches smaller than the chip,
# 192 = four workgroups per CU when 769..1024 runs make one launch, 0 = the default 3 x 256)
for n_runs, shape in [(1, 0), (1, 512), (64, 0), (64, 512), (256, 0), (256, 512), (768, 0), (1024, 0), (1024, 192), (4096, 0), (6144, 0), (16384, 0), (65536, 0)]:
    mod.set_workgroup_threads(shape)
    bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250102), **kw)
    mod.kernel_time(reset=True)
    t0 = time.perf_counter()
    costs, status = mod.batch_iterate(bid, 100)
    t1 = time.perf_counter()
    ms, n = mod.kernel_time()
    mod.batch_destroy(bid)
    out["points"].append({"batch": n_runs, "workgroup_threads": shape or 256, "it_per_s_wall": n_runs * 100 / (t1 - t0), "kernel_ms": ms / max(n, 1),
                          "it_per_s_kernel": n_runs * 100 / (ms / max(n, 1) * 1e-3), "runs_outside_joint_limits": int((status != 0).sum())})
print(json.dumps(out))
