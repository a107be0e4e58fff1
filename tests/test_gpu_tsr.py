"""TSR hard constraints (`con_tsr 'all ...'`, `everyn_tsr`) on the GPU against the
oracle's restatement of src/libcd/chomp.c:550-600 and src/orcdchomp_mod.cpp:1330-1657, through the
command layer (`createbatch ... con_tsr ...`), the way the reference's python layer issues them."""
import numpy as np
import pytest

import common
import or_cdchomp_amd
from or_cdchomp_amd import robots

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle_py as O
    O.build(ref=False)
    return O


def _start_frame(O, model, base, dofvals, link, tool):
    """pose pieces of the end effector at the start configuration (rotation, translation)"""
    rob = O.OraRobot(model)
    R, t, _, _ = rob.fk(base, dofvals)
    li = model.link_names.index(link)
    Rt = np.array(tool[3:])
    # tool rotation is the identity in these tests: the frame is the link's, moved by the tool offset
    assert np.allclose(Rt, [0, 0, 0, 1])
    return R[li], t[li] + R[li] @ np.array(tool[:3]), li


def _near_goals(n_runs, seed, spread=0.4):
    rng = np.random.default_rng(seed)
    return np.array(robots.WAM_START)[None, :] + spread * rng.uniform(-1, 1, size=(n_runs, 7))


def _unit_base():
    s2 = np.sqrt(0.5)
    return [-1.0, 0.0, 1.0, 0.0, s2, 0.0, s2]


def _setup(mod, base):
    model, _, dofvals, adofs = common.wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    from or_cdchomp_amd import scenes
    scenes.add_tabletop(mod)
    mod.SendCommand("computedistancefield kinbody table")
    return model, dofvals, adofs


@pytest.mark.parametrize("spec,link,tool", [("all link wam7", "wam7", [0, 0, 0, 0, 0, 0, 1]),
                                            ("all manipee arm", "handbase", [0, 0, 0.16, 0, 0, 0, 1]),
                                            ("all", "handbase", [0, 0, 0.16, 0, 0, 0, 1])])
def test_con_tsr_all_matches_oracle(oracle, spec, link, tool):
    """keep the end effector's height and two of its angles: three rows on every moving point"""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    Ree, tee, li = _start_frame(O, model, base, dofvals, link, tool)
    Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0], [0, 0], [-3, 3]]      # z, roll, pitch fixed
    tsr = robots.Tsr(T0w_R=Ree, T0w_d=tee, Bw=Bw)
    n_runs, n_points, n_iter = 4, 40, 25
    goals = _near_goals(n_runs, 3)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 con_tsr '%s' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, spec, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    T0w = O.pose_from_dR(tee, Ree)
    worst = 0.0
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, obs_factor=200.0))
        assert run.add_contsr(li, tool, T0w, [0, 0, 0, 0, 0, 0, 1], Bw) == 3
        before = max(np.abs(run.eval_contsr(0, run.traj()[i])[0]).max() for i in range(1, n_points - 1))
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[k] == 0
        err = common.rel_l2(traj[k], run.traj())
        worst = max(worst, err)
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        # and the constraint is doing its job: the straight line violates it, the result does not
        after = max(np.abs(run.eval_contsr(0, traj[k][i])[0]).max() for i in range(1, n_points - 1))
        assert before > 1e-2 and after < 0.02 * before, (before, after)
        run.destroy()
    assert worst <= 1e-6, worst


def test_everyn_tsr_with_con_tsr_and_momentum(oracle):
    """everyn_tsr (active manipulator) together with a con_tsr on a link, momentum on: two constraints
    per point, the reference's list order (the con_tsr's blocks first)"""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    tool = [0, 0, 0.16, 0, 0, 0, 1]
    Re, te, le = _start_frame(O, model, base, dofvals, "handbase", tool)
    Rl, tl, ll = _start_frame(O, model, base, dofvals, "wam4", [0, 0, 0, 0, 0, 0, 1])
    Bw_e = [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]      # hand height
    Bw_l = [[0, 0], [-1, 1], [-1, 1], [-3, 3], [-3, 3], [-3, 3]]      # elbow x
    tsr_e = robots.Tsr(T0w_R=Re, T0w_d=te, Bw=Bw_e)
    tsr_l = robots.Tsr(T0w_R=Rl, T0w_d=tl, Bw=Bw_l)
    n_runs, n_points, n_iter = 3, 30, 20
    goals = _near_goals(n_runs, 7, spread=0.3)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 use_momentum "
                              "everyn_tsr '%s' con_tsr 'all link wam4' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr_e.serialize(), tsr_l.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, use_momentum=1))
        run.add_contsr(le, tool, O.pose_from_dR(te, Re), [0, 0, 0, 0, 0, 0, 1], Bw_e)      # everyn_tsr is added first
        run.add_contsr(ll, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(tl, Rl), [0, 0, 0, 0, 0, 0, 1], Bw_l)
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[k] == 0
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        run.destroy()


@pytest.mark.parametrize("extra", ["", "momentum+con_tsr", "limits"])
def test_start_tsr_matches_oracle(oracle, extra):
    """`start_tsr` (reference src/orcdchomp_mod.cpp:1988-1992, 2316-2323, 2570-2576): the start point is a variable
    (m = n_points - 1, no start boundary in the metric, one-sided sphere velocity) held on a TSR: here the
    hand keeps its position while the arm's start configuration moves.  Second case: momentum and a
    con_tsr on every point on top of it (three constraints' worth of blocks in the reference's list order).
    Third case: goals near the joint limits, so that the joint-limit rounds run on the metric without a
    start boundary (cyclic reduction instead of the closed forms of the default metric)."""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    tool = [0, 0, 0.16, 0, 0, 0, 1]
    Ree, tee, li = _start_frame(O, model, base, dofvals, "handbase", tool)
    Bw_s = [[0, 0], [0, 0], [0, 0], [-3, 3], [-3, 3], [-3, 3]]      # hand position fixed, orientation free
    tsr_s = robots.Tsr(T0w_R=Ree, T0w_d=tee, Bw=Bw_s)
    Rl, tl, ll = _start_frame(O, model, base, dofvals, "wam4", [0, 0, 0, 0, 0, 0, 1])
    Bw_l = [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]     # elbow height
    tsr_l = robots.Tsr(T0w_R=Rl, T0w_d=tl, Bw=Bw_l)
    n_runs, n_points, n_iter = 4, 30, 20
    goals = _near_goals(n_runs, 11, spread=0.3)
    if extra == "limits":
        n_runs, n_points, n_iter = 12, 40, 40
        goals = common.wam_goals(n_runs, seed=20250101)
    with_con = (extra == "momentum+con_tsr")
    more = ("use_momentum con_tsr 'all link wam4' '%s'" % tsr_l.serialize()) if with_con else ""
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 start_tsr '%s' %s"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr_s.serialize(), more)))
    seed = mod.batch_gettraj(bid)
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    assert traj.shape == (n_runs, n_points, 7)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    T0w = O.pose_from_dR(tee, Ree)
    limadjs = 0
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, obs_factor=200.0, use_momentum=1 if with_con else 0,
                                        start_tsr=(li, tool, T0w, [0, 0, 0, 0, 0, 0, 1], Bw_s)))
        assert run.m == n_points - 1
        if with_con:
            run.add_contsr(ll, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(tl, Rl), [0, 0, 0, 0, 0, 0, 1], Bw_l)
        assert np.array_equal(seed[k], run.traj())
        st, oc = run.iterate(n_iter)
        limadjs += run.chomp().last_num_limadjs
        assert st == status[k]
        if st != 0:
            run.destroy()
            continue
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6, common.rel_l2(traj[k], run.traj())
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        # the start configuration moved, the goal did not, and the hand is where it was
        assert np.abs(traj[k][0] - seed[k][0]).max() > 1e-3 and np.array_equal(traj[k][-1], seed[k][-1])
        _, t2, _, _ = rob.fk(base, [*traj[k][0], *dofvals[7:]])
        R2 = rob.fk(base, [*traj[k][0], *dofvals[7:]])[0]
        # (the joint-limit rounds come after the constraint step and pull the start off the TSR again)
        assert extra == "limits" or np.abs(t2[li] + R2[li] @ np.array(tool[:3]) - tee).max() < 1e-4
        run.destroy()
    if extra == "limits":
        assert limadjs > 0, "the workload is expected to make joint-limit rounds"


def test_start_tsr_many_sphere_robot(oracle):
    """`start_tsr` on the kernel variant for robots with more than 16 spheres (30-dof tree, 60 spheres): the
    tip of the left arm keeps its position while the start configuration moves"""
    O = oracle
    from or_cdchomp_amd import scenes
    mod = or_cdchomp_amd.Module(0)
    model = robots.tree30()
    base = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    rng = np.random.default_rng(3)
    dofvals = rng.uniform(-0.3, 0.3, size=model.n_dof)
    adofs = list(range(model.n_dof))
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    li = model.link_names.index("L13")
    tool = [0, 0, 0.15, 0, 0, 0, 1]
    mod.add_manipulator(model.name, "left", li, tool)
    grids, poses = [], []
    for name, (boxes, pose) in scenes.random_boxes(np.random.default_rng(20250104)).items():
        mod.add_kinbody_boxes(name, boxes, transform=pose)
        mod.SendCommand("computedistancefield kinbody %s cube_extent 0.02 aabb_padding 0.15" % name)
        data, lengths, gpose = mod.get_sdf(name)
        grids.append(O.OraGrid(data, lengths))
        out = np.zeros(7)
        O.lib().ora_kin_pose_compose(O.dp(O.f64(pose)), O.dp(O.f64(gpose)), O.dp(out))
        poses.append(out)
    rob = O.OraRobot(model)
    R, t, _, _ = rob.fk(base, dofvals)
    tee = t[li] + R[li] @ np.array(tool[:3])
    Bw = [[0, 0], [0, 0], [0, 0], [-3, 3], [-3, 3], [-3, 3]]
    tsr = robots.Tsr(T0w_R=R[li], T0w_d=tee, Bw=Bw)
    n_runs, n_points, n_iter = 2, 24, 10
    goals = dofvals[None, :] + rng.uniform(-0.4, 0.4, size=(n_runs, model.n_dof))
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 200 obs_factor 100 start_tsr '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize())))
    seed = mod.batch_gettraj(bid)
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], grids, poses,
                       O.default_params(n_points=n_points, lambda_=200.0, obs_factor=100.0,
                                        start_tsr=(li, tool, O.pose_from_dR(tee, R[li]), [0, 0, 0, 0, 0, 0, 1], Bw)))
        assert run.Sa == 60 and run.m == n_points - 1
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[k] == 0
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6, common.rel_l2(traj[k], run.traj())
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        assert np.abs(traj[k][0] - seed[k][0]).max() > 1e-4
        run.destroy()


@pytest.mark.parametrize("where", ["lds", "global"])
def test_con_tsr_many_sphere_robot_trajectory_in_global_memory(oracle, monkeypatch, where):
    """`con_tsr` on a robot with more than 16 spheres, whose kernels may iterate the trajectory in global memory (a long
    trajectory of many dofs does not fit the LDS beside its tiles): the constraint step reads and moves it where it lives.
    ORC_T_LDS=0 / ORC_G_LDS=0 ask the planner for that layout on a run short enough for the oracle."""
    O = oracle
    if where == "global":
        monkeypatch.setenv("ORC_T_LDS", "0")
        monkeypatch.setenv("ORC_G_LDS", "0")
    mod = or_cdchomp_amd.Module(0)
    model = robots.tree30()
    base = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    rng = np.random.default_rng(5)
    dofvals = rng.uniform(-0.3, 0.3, size=model.n_dof)
    adofs = list(range(model.n_dof))
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    li = model.link_names.index("L13")
    tool = [0, 0, 0.15, 0, 0, 0, 1]
    mod.add_manipulator(model.name, "left", li, tool)
    from or_cdchomp_amd import scenes
    grids, poses = [], []
    for name, (boxes, pose) in scenes.random_boxes(np.random.default_rng(20250104)).items():
        mod.add_kinbody_boxes(name, boxes, transform=pose)
        mod.SendCommand("computedistancefield kinbody %s cube_extent 0.02 aabb_padding 0.15" % name)
        data, lengths, gpose = mod.get_sdf(name)
        grids.append(O.OraGrid(data, lengths))
        out = np.zeros(7)
        O.lib().ora_kin_pose_compose(O.dp(O.f64(pose)), O.dp(O.f64(gpose)), O.dp(out))
        poses.append(out)
    rob = O.OraRobot(model)
    R, t, _, _ = rob.fk(base, dofvals)
    tee = t[li] + R[li] @ np.array(tool[:3])
    Bw = [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]          # the tool keeps its height in its own start frame
    tsr = robots.Tsr(T0w_R=R[li], T0w_d=tee, Bw=Bw)
    n_runs, n_points, n_iter = 3, 30, 12
    goals = dofvals[None, :] + rng.uniform(-0.3, 0.3, size=(n_runs, model.n_dof))
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 200 obs_factor 100 "
                              "con_tsr 'all manipee left' '%s'" % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], grids, poses, O.default_params(n_points=n_points, lambda_=200.0, obs_factor=100.0))
        assert run.add_contsr(li, tool, O.pose_from_dR(tee, R[li]), [0, 0, 0, 0, 0, 0, 1], Bw) == 1
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[k] == 0
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6, common.rel_l2(traj[k], run.traj())
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        run.destroy()


def test_start_tsr_argument_errors():
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, _unit_base())
    tsr = robots.Tsr(Bw=[[0, 0]] * 3 + [[-3, 3]] * 3)
    with pytest.raises(RuntimeError, match="floating_base and start_tsr together is not yet implemented"):
        mod.SendCommand("create robot %s adofgoal '%s' basegoal '0 0 0 0 0 0 1' floating_base start_tsr '%s'"
                        % (model.name, " ".join(map(str, robots.WAM_GOAL)), tsr.serialize()))
    with pytest.raises(RuntimeError, match="Cannot parse start_tsr TSR"):
        mod.SendCommand("create robot %s adofgoal '%s' start_tsr '0 NULL 1 2 3'" % (model.name, " ".join(map(str, robots.WAM_GOAL))))


def test_structured_and_dense_constraint_solves_agree(oracle, monkeypatch):
    """The constraint step solves the block-tridiagonal KKT system point by point (csrc/tsr.h); the dense
    J Ainv J^T + LU of the reference's formulation stays as the fallback (ORC_TSR_DENSE=1 selects it).  Both on
    the same batch; and with a constraint given twice (the reference's dgesv finds the system singular, prints
    "constraint inversion error!" and goes on with the unsolved right-hand side, src/libcd/chomp.c:579-590):
    the point-by-point solve meets a zero pivot and hands the iteration to the dense path, so the two builds
    of the step still agree.  (Whether LAPACK itself meets an exact zero there is a matter of rounding: that
    case is not compared with the oracle.)"""
    O = oracle
    base = _unit_base()
    out = {}
    tool = [0, 0, 0.16, 0, 0, 0, 1]
    Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0], [-3, 3], [-3, 3]]
    n_runs, n_points, n_iter = 6, 40, 15
    goals = _near_goals(n_runs, 5)
    for twice in (False, True):
        for mode in ("structured", "dense"):
            if mode == "dense":
                monkeypatch.setenv("ORC_TSR_DENSE", "1")
            else:
                monkeypatch.delenv("ORC_TSR_DENSE", raising=False)
            mod = or_cdchomp_amd.Module(0)
            model, dofvals, adofs = _setup(mod, base)
            Ree, tee, li = _start_frame(O, model, base, dofvals, "handbase", tool)
            tsr = robots.Tsr(T0w_R=Ree, T0w_d=tee, Bw=Bw)
            con = "con_tsr 'all' '%s'" % tsr.serialize()
            bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 %s"
                                      % (model.name, n_runs, goals.ctypes.data, n_points, con + " " + con if twice else con)))
            costs, status = mod.batch_iterate(bid, 3 if twice else n_iter)
            out[mode] = (mod.batch_gettraj(bid), costs, status)
            mod.batch_destroy(bid)
        assert np.array_equal(out["structured"][2], out["dense"][2])
        for k in range(n_runs):
            if out["dense"][2][k] == 0:
                assert common.rel_l2(out["structured"][0][k], out["dense"][0][k]) <= 1e-10
                assert np.allclose(out["structured"][1][k], out["dense"][1][k], rtol=1e-10, atol=0)
        if not twice:
            assert (out["dense"][2] == 0).all()


def test_con_tsr_floating_base(oracle):
    """floating base: the base pose columns of the constraint Jacobian come from cd_spatial_pose_jac"""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    tool = [0, 0, 0, 0, 0, 0, 1]
    Re, te, le = _start_frame(O, model, base, dofvals, "wam7", tool)
    Bw = [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]
    tsr = robots.Tsr(T0w_R=Re, T0w_d=te, Bw=Bw)
    n_runs, n_points, n_iter = 2, 24, 12
    goals = _near_goals(n_runs, 11, spread=0.25)
    rng = np.random.default_rng(5)
    basegoals = np.tile(np.array(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.1, 0.1, size=(n_runs, 3))
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x basegoals 0x%x floating_base n_points %d lambda 100 "
                              "con_tsr 'all link wam7' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, basegoals.ctypes.data, n_points, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, floating_base=1), basegoal=basegoals[k])
        run.add_contsr(le, tool, O.pose_from_dR(te, Re), [0, 0, 0, 0, 0, 0, 1], Bw)
        st, oc = run.iterate(n_iter)
        assert st == status[k]
        if st == 0:
            assert common.rel_l2(traj[k], run.traj()) <= 1e-6
            assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        run.destroy()


def test_tsr_argument_errors():
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, _unit_base())
    tsr = robots.Tsr().serialize()
    with pytest.raises(RuntimeError, match="con_tsr first arg must be start, end, or all!"):
        mod.SendCommand("create robot %s adofgoal '0 0 0 0 0 0 0' con_tsr 'start' '%s'" % (model.name, tsr))
    with pytest.raises(RuntimeError, match="con_tsr link not found!"):
        mod.SendCommand("create robot %s adofgoal '0 0 0 0 0 0 0' con_tsr 'all link nope' '%s'" % (model.name, tsr))
    with pytest.raises(RuntimeError, match="con_tsr manip not found!"):
        mod.SendCommand("create robot %s adofgoal '0 0 0 0 0 0 0' con_tsr 'all manipee nope' '%s'" % (model.name, tsr))
    with pytest.raises(RuntimeError, match="Cannot parse constraint TSR!"):
        mod.SendCommand("create robot %s adofgoal '0 0 0 0 0 0 0' con_tsr 'all' '0 NULL 1 2 3'" % model.name)
    with pytest.raises(RuntimeError, match="You must pass robot before any con_tsrs!"):
        mod.SendCommand("create con_tsr 'all' '%s' robot %s adofgoal '0 0 0 0 0 0 0'" % (tsr, model.name))


def test_con_tsr_full_trajectory_length(oracle):
    """BASELINE configs[1] shapes with a constraint: 100 waypoints, 64 runs, two constrained rows per moving
    point (a 196 x 196 system per run and iteration), three runs of the batch against the oracle"""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    tool = [0, 0, 0, 0, 0, 0, 1]
    Re, te, le = _start_frame(O, model, base, dofvals, "wam7", tool)
    Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0], [-3, 3], [-3, 3]]      # height and roll
    tsr = robots.Tsr(T0w_R=Re, T0w_d=te, Bw=Bw)
    n_runs, n_points, n_iter = 64, 100, 30
    goals = _near_goals(n_runs, 21, spread=0.35)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 500 con_tsr 'all link wam7' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    for k in (0, 31, 63):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, obs_factor=500.0))
        assert run.add_contsr(le, tool, O.pose_from_dR(te, Re), [0, 0, 0, 0, 0, 0, 1], Bw) == 2
        st, oc = run.iterate(n_iter)
        assert st == status[k]
        if st == 0:
            assert common.rel_l2(traj[k], run.traj()) <= 1e-6
            assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        run.destroy()
    # a subset on its own: bit for bit (the constraint workspace is per run)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 500 con_tsr 'all link wam7' '%s'"
                              % (model.name, 8, np.ascontiguousarray(goals[8:16]).ctypes.data, n_points, tsr.serialize())))
    mod.batch_iterate(bid, n_iter)
    sub = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    assert np.array_equal(sub, traj[8:16])


def test_con_tsr_on_a_link_no_active_joint_moves(oracle):
    """a constraint nothing can satisfy or violate: the rows of a link the active joints do not move have a zero
    Jacobian, J Ainv J^T is the zero matrix, the reference's dgesv reports it singular ("constraint inversion
    error!", src/libcd/chomp.c:579-590), leaves h as it is and pushes J^T h = 0 to the trajectory -- the run goes on as
    if unconstrained.  The structured elimination meets a zero pivot and hands over to the dense path, which does the same."""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    Rl, tl, ll = _start_frame(O, model, base, dofvals, "wam0", [0, 0, 0, 0, 0, 0, 1])
    Bw = [[0, 0], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]
    tsr = robots.Tsr(T0w_R=Rl, T0w_d=tl + np.array([0.05, 0.0, -0.02]), Bw=Bw)      # and the link is not where the TSR wants it
    n_runs, n_points, n_iter = 3, 30, 8
    goals = _near_goals(n_runs, 11)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 con_tsr 'all link wam0' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    bid = mod.batch_create(model.name, goals, n_points=n_points, lambda_=100.0, obs_factor=200.0)
    costs_free, _ = mod.batch_iterate(bid, n_iter)
    traj_free = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, obs_factor=200.0))
        assert run.add_contsr(ll, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(tl + np.array([0.05, 0.0, -0.02]), Rl), [0, 0, 0, 0, 0, 0, 1], Bw) == 2
        h, J = run.eval_contsr(0, run.traj()[5])
        assert np.all(J == 0.0) and np.abs(h).max() > 0.01
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[k] == 0
        assert np.all(np.isfinite(traj[k]))
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        assert common.rel_l2(traj[k], traj_free[k]) <= 1e-9          # and that is the unconstrained run
        run.destroy()


def test_the_same_con_tsr_twice(oracle):
    """the same constraint passed twice: J Ainv J^T has every row twice, the elimination meets an exactly zero pivot, the
    reference's dgesv reports "constraint inversion error!" and leaves h as it is (src/libcd/chomp.c:579-590; dgetrs is
    never called), and h -- the constraint values, not multipliers -- goes through J^T and Ainv to the trajectory.
    Deterministic nonsense, and the same nonsense here: the structured elimination hands over to the dense path."""
    O = oracle
    base = _unit_base()
    mod = or_cdchomp_amd.Module(0)
    model, dofvals, adofs = _setup(mod, base)
    Ree, tee, li = _start_frame(O, model, base, dofvals, "wam7", [0, 0, 0, 0, 0, 0, 1])
    Bw = [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]
    tsr = robots.Tsr(T0w_R=Ree, T0w_d=tee, Bw=Bw)
    n_runs, n_points, n_iter = 3, 24, 4
    goals = _near_goals(n_runs, 5, spread=0.2)
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 "
                              "con_tsr 'all link wam7' '%s' con_tsr 'all link wam7' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize(), tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    T0w = O.pose_from_dR(tee, Ree)
    for k in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]],
                       O.default_params(n_points=n_points, lambda_=100.0, obs_factor=200.0))
        run.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], T0w, [0, 0, 0, 0, 0, 0, 1], Bw)
        run.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], T0w, [0, 0, 0, 0, 0, 0, 1], Bw)
        st, oc = run.iterate(n_iter)
        assert st == status[k]
        assert np.all(np.isfinite(traj[k])) and np.all(np.isfinite(run.traj()))
        assert common.rel_l2(traj[k], run.traj()) <= 1e-6, common.rel_l2(traj[k], run.traj())
        assert np.allclose(costs[k], oc, rtol=1e-6, atol=0)
        run.destroy()


@pytest.mark.parametrize("n_points", [400, 640])
def test_con_tsr_long_trajectory_on_an_overlapping_module(oracle, n_points):
    """round-5 advisor: on a module whose launches overlap (orc_set_num_streams >= 2) the planner prefers 128-thread
    workgroups for constrained runs; a long trajectory has no such plan (40 KB of LDS at four per CU) and must be planned
    like any other run instead of failing with "run does not fit the LDS of one CU!".  Same trajectories as streams = 0."""
    O = oracle
    base = _unit_base()
    tool = [0, 0, 0, 0, 0, 0, 1]
    n_runs, n_iter = 6, 4
    goals = np.ascontiguousarray(_near_goals(n_runs, 77, spread=0.3))
    out = {}
    for streams in (0, 2):
        mod = or_cdchomp_amd.Module(0)
        model, dofvals, adofs = _setup(mod, base)
        Re, te, le = _start_frame(O, model, base, dofvals, "wam7", tool)
        Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0], [-3, 3], [-3, 3]]
        tsr = robots.Tsr(T0w_R=Re, T0w_d=te, Bw=Bw)
        mod.set_num_streams(streams)
        bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 100 obs_factor 200 con_tsr 'all link wam7' '%s'"
                                  % (model.name, n_runs, goals.ctypes.data, n_points, tsr.serialize())))
        costs, status = mod.batch_iterate(bid, n_iter)
        out[streams] = (mod.batch_gettraj(bid), costs, status)
        mod.batch_destroy(bid)
        mod.close()
    assert (out[0][2] == 0).all() and (out[2][2] == 0).all()
    for k in range(n_runs):
        assert common.rel_l2(out[2][0][k], out[0][0][k]) <= 1e-9
    assert np.allclose(out[2][1], out[0][1], rtol=1e-9, atol=0)
    # and one run against the oracle
    prob = common.tabletop_problem(O)
    run = O.OraRun(O.OraRobot(model), base, dofvals, adofs, goals[1], [prob["sdf"]], [prob["pose"]],
                   O.default_params(n_points=n_points, lambda_=100.0, obs_factor=200.0))
    assert run.add_contsr(le, tool, O.pose_from_dR(te, Re), [0, 0, 0, 0, 0, 0, 1], Bw) == 2
    st, oc = run.iterate(n_iter)
    assert st == 0
    assert common.rel_l2(out[2][0][1], run.traj()) <= 1e-6
    assert np.allclose(out[2][1][1], oc, rtol=1e-6, atol=0)
    run.destroy()


def _chain(n_dof):
    """a serial chain of n_dof revolute joints, alternating z / y / x axes with a bend in between, 0.12 m links, one sphere per link
    (at most 16 active spheres: the 16-lane cost pass)"""
    r = robots.RobotModel("chain%d" % n_dof)
    r.add_link("base")
    prev = "base"
    axes = ((0, 0, 1), (0, 1, 0), (1, 0, 0))
    for i in range(n_dof):
        nm = "l%d" % i
        r.add_link(nm, prev, (0.02 if i % 2 else 0.0, 0.0, 0.12), quat=robots.quat_from_axis_angle((1, 0.3 * i, 0.2), 0.25 * ((i % 3) - 1)),
                   joint=robots.JOINT_REVOLUTE, axis=axes[i % 3], limits=(-2.5, 2.5))
        if i < 16:
            r.add_sphere(nm, (0, 0, 0.06), 0.04)
        prev = nm
    return r


@pytest.mark.parametrize("n_dof,k", [(5, 3), (9, 3), (12, 3), (13, 3), (14, 3), (17, 3), (20, 3), (21, 3), (22, 3)])
def test_con_tsr_every_register_shape_of_the_elimination(oracle, n_dof, k):
    """the block of a point has N = n + k rows: rows of 16 lanes with 2 (N <= 8, the augmented form when 2 n + k + 1 <= 16), 3 (<= 12) and
    4 (<= 15) registers, rows of 32 lanes with 8 (<= 16), 10 (<= 20) and 12 (<= 24) registers, the LDS form beyond: a chain of n
    revolute joints with k constrained rows on its last link walks them all (csrc/tsr.h phase_tsr; src/libcd/chomp.c:553-600)"""
    O = oracle
    model = _chain(n_dof)
    base = [0.2, -0.1, 0.9, 0, 0, 0, 1]
    rng = np.random.default_rng(100 + n_dof)
    dofvals = 0.4 * rng.uniform(-1, 1, size=n_dof)
    adofs = list(range(n_dof))
    mod = or_cdchomp_amd.Module(0)
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    from or_cdchomp_amd import scenes
    scenes.add_tabletop(mod)
    mod.SendCommand("computedistancefield kinbody table")
    rob = O.OraRobot(model)
    R, t, _, _ = rob.fk(base, dofvals)
    li = model.link_names.index("l%d" % (n_dof - 1))
    Bw = [[-1, 1], [0, 0], [0, 0], [0, 0], [-3, 3], [-3, 3]] if k == 3 else [[-1, 1], [-1, 1], [0, 0], [-3, 3], [-3, 3], [-3, 3]]
    tsr = robots.Tsr(T0w_R=R[li], T0w_d=t[li], Bw=Bw)
    n_runs, n_points, n_iter = 3, 24, 8
    goals = np.ascontiguousarray(dofvals[None, :] + 0.25 * rng.uniform(-1, 1, size=(n_runs, n_dof)))
    bid = int(mod.SendCommand("createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda 200 obs_factor 100 con_tsr 'all link l%d' '%s'"
                              % (model.name, n_runs, goals.ctypes.data, n_points, n_dof - 1, tsr.serialize())))
    costs, status = mod.batch_iterate(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    prob = common.tabletop_problem(O)
    worst = 0.0
    for r in range(n_runs):
        run = O.OraRun(rob, base, dofvals, adofs, goals[r], [prob["sdf"]], [prob["pose"]], O.default_params(n_points=n_points, lambda_=200.0, obs_factor=100.0))
        assert run.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], Bw) == k
        st, oc = run.iterate(n_iter)
        assert st == 0 and status[r] == 0, (n_dof, r, st, status[r])
        worst = max(worst, common.rel_l2(traj[r], run.traj()))
        assert np.allclose(costs[r], oc, rtol=1e-6, atol=1e-12), (n_dof, r, costs[r], oc)
        run.destroy()
    assert worst <= 1e-6, (n_dof, worst)
