"""diagnostic: per-run parity of the WAM holding the four-sphere box, with the oracle's own one-ulp amplification"""
import sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common, or_cdchomp_amd
from or_cdchomp_amd import robots
from oracle import oracle_py as O
import test_gpu_grabbed as T
O.build(ref=False)
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
_, base, dofvals, adofs = common.wam_state()
prob = common.tabletop_problem(O)
box_pose = T._hand_pose(model, base, dofvals, (-0.04, -0.05, 0.15))
mod.add_kinbody_boxes("box", [([0.045, 0.045, 0.01, 0, 0, 0, 1], [0.08, 0.08, 0.04])], transform=box_pose)
mod.set_kinbody_spheres("box", T.BOX_POS, T.BOX_RAD)
hand = model.link_names.index("handbase")
mod.grab(model.name, "box", hand)
goals = common.wam_goals(12, seed=42)
for n_iter in (1, 5, 20, 60):
    bid = mod.batch_create(model.name, goals, **T.KW)
    costs, st = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid); mod.batch_destroy(bid)
    g = [(hand, box_pose, T.BOX_POS, T.BOX_RAD)]
    rob = O.OraRobot(model, grabbed=g)
    ot, oc, ost, _ = O.batch_run(rob, base, dofvals, adofs, goals, [prob["sdf"]], [prob["pose"]], O.default_params(**T.KW), n_iter)
    pt, pc, pst, _ = O.batch_run(rob, base, dofvals, adofs, goals * (1 + 2.0**-52), [prob["sdf"]], [prob["pose"]], O.default_params(**T.KW), n_iter)
    err = [common.rel_l2(traj[k], ot[k]) for k in range(12)]
    amp = [common.rel_l2(pt[k], ot[k]) for k in range(12)]
    print("n_iter", n_iter, "status", st.tolist(), ost.tolist())
    print("  hip vs oracle ", " ".join("%.1e" % e for e in err))
    print("  oracle +1ulp  ", " ".join("%.1e" % e for e in amp))
