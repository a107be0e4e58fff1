"""it/s and kernel family of config 2's WAM while it HOLDS something (reference src/orcdchomp_mod.cpp:2168-2300): a one-sphere
body (16 active spheres: still a 16-lane row) and a four-sphere body (19 active: the many-sphere pass).
   python scripts/grabbed_rate.py [n_runs=1024]"""
import os, sys, time
os.environ.setdefault("ORC_DEBUG_PLAN", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common, or_cdchomp_amd
from or_cdchomp_amd import robots
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BODIES = {"nothing": None,
          "cup (1 sphere)": ([[0.0, 0.0, 0.03]], [0.05], (0.0, 0.0, 0.17)),
          "box (4 spheres)": ([[0.0, 0.0, 0.0], [0.09, 0.0, 0.0], [0.0, 0.09, 0.02], [0.09, 0.09, 0.02]], [0.05, 0.045, 0.04, 0.05], (-0.04, -0.05, 0.15))}
for name, body in BODIES.items():
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    _, base, dofvals, _ = common.wam_state()
    if body:
        R, t = model.link_frames(base, dofvals)
        li = model.link_names.index("handbase")
        pose = list(t[li] + R[li] @ np.asarray(body[2])) + list(robots.quat_from_axis_angle((0.3, -0.5, 0.8), 0.7))
        mod.add_kinbody_boxes("held", [([0, 0, 0, 0, 0, 0, 1], [0.02, 0.02, 0.02])], transform=pose)
        mod.set_kinbody_spheres("held", body[0], body[1])
        mod.grab(model.name, "held", li)
    rates = []
    for step in range(4):
        bid = mod.batch_create(model.name, common.wam_goals(n_runs, seed=20250101 + step), **common.CONFIG2_KW)
        mod.kernel_time(reset=True)
        t0 = time.perf_counter(); costs, status = mod.batch_iterate(bid, 100); t1 = time.perf_counter()
        ms, _ = mod.kernel_time()
        made = int(mod.batch_iterations_done(bid).sum())
        rates.append(made / (ms * 1e-3))
        mod.batch_destroy(bid)
    print("WAM holding %-16s: %d runs x 100 iterations, one launch at a time: %.3g it/s (kernel time, best of 3 after a warm-up), runs outside their limits %d" % (
        name, n_runs, max(rates[1:]), int((status != 0).sum())))
    mod.close()
