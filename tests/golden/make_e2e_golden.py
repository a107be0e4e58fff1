#!/usr/bin/env python3
"""Generates tests/golden/e2e_wam_config1.npz: BASELINE.json configs[0] (WAM, the demo's start, the
synthetic goal and tabletop of SURVEY.md 8d config 1, n_points 101, lambda 100, obs_factor 500)
run by the ORACLE for 1, 10 and 100 iterations.  This is a regression anchor for the oracle and the
HIP path (the oracle's own pinning status is stated in oracle/oracle.h and DESIGN.md section 4: its
CHOMP part follows the reference's source text, which cannot be built here).

   python tests/golden/make_e2e_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O  # noqa: E402
import common  # noqa: E402
from or_cdchomp_amd import robots  # noqa: E402

O.build(ref=False)
model, base, dofvals, adofs = common.wam_state()
prob = common.tabletop_problem(O)
goal = np.asarray(robots.WAM_GOAL, dtype=np.float64)
kw = dict(n_points=101, lambda_=100.0, obs_factor=500.0)
out = {"goal": goal, "start": np.asarray(dofvals[:7]), "base_pose": np.asarray(base),
       "sdf_sizes": np.asarray(prob["sizes"]), "sdf_lengths": np.asarray(prob["lengths"]), "sdf_pose": np.asarray(prob["pose"])}
rob = O.OraRobot(model)
for n_iter in (1, 10, 100):
    run = O.OraRun(rob, base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]], O.default_params(**kw))
    if n_iter == 1:
        out["seed_traj"] = run.traj().copy()
    st, costs = run.iterate(n_iter)
    assert st == 0
    out["traj_%d" % n_iter] = run.traj().copy()
    out["costs_%d" % n_iter] = np.asarray(costs)
    run.destroy()
np.savez_compressed(os.path.join(HERE, "e2e_wam_config1.npz"), **out)
print({k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})
print("costs after 100 iterations:", out["costs_100"])
