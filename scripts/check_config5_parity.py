import sys; sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests")
import numpy as np, common, or_cdchomp_amd
from oracle import oracle_py as O
from or_cdchomp_amd import robots, scenes
O.build(ref=False)
mod = or_cdchomp_amd.Module(0)
model = robots.tree30()
base = [0.0]*6 + [1.0]; dofvals = np.zeros(model.n_dof); adofs = list(range(model.n_dof))
mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
rng = np.random.default_rng(20250104)
grids, poses = [], []
for name, (boxes, pose) in scenes.random_boxes(rng).items():
    mod.add_kinbody_boxes(name, boxes, transform=pose)
    mod.SendCommand("computedistancefield kinbody %s cube_extent 0.02 aabb_padding 0.15" % name)
    data, lengths, gpose = mod.get_sdf(name)
    grids.append(O.OraGrid(data, lengths))
    out = np.zeros(7); O.lib().ora_kin_pose_compose(O.dp(O.f64(pose)), O.dp(O.f64(gpose)), O.dp(out)); poses.append(out)
goals = np.random.default_rng(5).uniform(-0.8, 0.8, size=(2, model.n_dof))
kw = dict(n_points=200, lambda_=200.0, obs_factor=100.0)
rob = O.OraRobot(model)
for prec in (64, 32):
    bid = mod.batch_create(model.name, goals, precision=prec, **kw)
    costs, status = mod.batch_iterate(bid, 30)
    traj = mod.batch_gettraj(bid); mod.batch_destroy(bid)
    errs = []
    for k in range(2):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], grids, poses, O.default_params(**kw))
        st, oc = run.iterate(30)
        errs.append(common.rel_l2(traj[k], run.traj())); run.destroy()
    print("precision %d, n_points 200, 30 iterations: worst rel L2 %.3e, costs %s vs oracle %s" % (prec, max(errs), costs[1], oc))
