"""-m gpu: seeded random robots against the oracle.

The kernel is compiled in many shapes (16 / 32 / 64 lanes per waypoint, chains with their J^T suffix scan, trees with
ranges or masks, the FK walk cut in two, fixed or floating base, spheres placed on the row, inactive spheres, the
matrix-core nomination of pairs in fp32): which of them a robot gets is decided at `create` from its description.
The named configurations pin a handful of robots; this test draws robots nobody tuned for -- random topology,
joint types, axes, fixed transforms, sphere counts, active dofs and run parameters -- and asks the same of each:
the trajectory and the costs of the reference's arithmetic (oracle/ora_run.c, src/orcdchomp_mod.cpp:968-1327,
src/libcd/chomp.c:430-683) to 1e-6 (fp64) / 1e-3 (fp32, which the reference does not have)."""
import math
import os

import numpy as np
import pytest

import common
from or_cdchomp_amd import robots, scenes

pytestmark = pytest.mark.gpu


def _random_quat(rng, max_angle):
    axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
    return robots.quat_from_axis_angle(tuple(axis), float(rng.uniform(-max_angle, max_angle)))


def random_robot(seed):
    """returns (model, description) -- a robot of 2..26 moving joints"""
    rng = np.random.default_rng(1000 + seed)
    kind = ("chain", "fork", "tree")[int(rng.integers(0, 3))]
    n_joints = int(rng.integers(2, 27 if kind != "chain" else 17))
    many = bool(rng.uniform() < 0.15)
    model = robots.RobotModel("rnd%d" % seed)
    model.add_link("base")
    names = ["base"]
    cut = int(rng.integers(1, max(2, n_joints // 2)))               # fork: a chain of `cut` joints, then two branches
    tips = []
    for j in range(n_joints):
        if kind == "chain":
            parent = names[-1]
        elif kind == "fork":
            if j <= cut:
                parent = names[-1]
            else:
                if len(tips) < 2:
                    tips = [names[cut], names[cut]]
                b = int(rng.integers(0, 2))
                parent = tips[b]
        else:
            parent = names[int(rng.integers(max(0, len(names) - 4), len(names)))]
        # now and then a fixed link between two joints
        if rng.uniform() < 0.12:
            fx = "f%d" % j
            model.add_link(fx, parent, tuple(rng.uniform(-0.05, 0.05, size=3)), _random_quat(rng, 0.6), joint=robots.JOINT_FIXED)
            if rng.uniform() < 0.5:
                model.add_sphere(fx, tuple(rng.uniform(-0.02, 0.02, size=3)), float(rng.uniform(0.03, 0.05)))
            parent = fx
        nm = "j%d" % j
        prismatic = rng.uniform() < 0.15
        axis = rng.normal(size=3)
        if rng.uniform() < 0.5:
            axis = np.eye(3)[int(rng.integers(0, 3))]
        limits = (-0.25, 0.25) if prismatic else ((-2.2, 2.2) if rng.uniform() < 0.8 else None)
        model.add_link(nm, parent, tuple(rng.uniform(-0.04, 0.04, size=2)) + (float(rng.uniform(0.08, 0.2)),),
                       _random_quat(rng, 0.5) if rng.uniform() < 0.4 else (0, 0, 0, 1),
                       joint=robots.JOINT_PRISMATIC if prismatic else robots.JOINT_REVOLUTE, axis=tuple(axis), limits=limits)
        names.append(nm)
        if kind == "fork" and j > cut:
            tips[b] = nm
    # spheres: 0..3 per moving link, at most 60 in all, at least one on the last link
    budget = 60 - len(model.spheres)
    for nm in names[1:]:
        k = int(rng.integers(0, 4)) if n_joints <= 16 else int(rng.integers(0, 3))
        if many:
            k = int(rng.integers(2, 4))                                  # more than 32 spheres: a wavefront per waypoint
        if nm == names[-1]:
            k = max(k, 1)
        for _ in range(min(k, budget)):
            model.add_sphere(nm, tuple(rng.uniform(-0.03, 0.03, size=2)) + (float(rng.uniform(0.0, 0.12)),), float(rng.uniform(0.03, 0.07)))
            budget -= 1
    if rng.uniform() < 0.5:
        model.add_sphere("base", (0.0, 0.0, 0.05), 0.08)               # never moves: an inactive sphere
    return model, "%s of %d joints, %d spheres" % (kind, n_joints, len(model.spheres))


def _scene(mod, oracle, which):
    if which == "table":
        scenes.add_tabletop(mod)
        mod.SendCommand("computedistancefield kinbody table")
        prob = common.tabletop_problem(oracle)
        return [prob["sdf"]], [prob["pose"]]
    rng = np.random.default_rng(20250104)
    grids, poses = [], []
    for name, (boxes, pose) in scenes.random_boxes(rng, n_bodies=which).items():
        mod.add_kinbody_boxes(name, boxes, transform=pose)
        mod.SendCommand("computedistancefield kinbody %s cube_extent 0.02 aabb_padding 0.15" % name)
        data, lengths, gpose = mod.get_sdf(name)
        grids.append(oracle.OraGrid(data, lengths))
        out = np.zeros(7)
        oracle.lib().ora_kin_pose_compose(oracle.dp(oracle.f64(pose)), oracle.dp(oracle.f64(gpose)), oracle.dp(out))
        poses.append(out)
    return grids, poses


SEEDS = list(range(int(os.environ.get("ORC_RANDOM_ROBOTS", "24"))))       # more of them: ORC_RANDOM_ROBOTS=400 pytest ...


@pytest.mark.parametrize("seed", SEEDS)
def test_random_robot_matches_oracle(oracle, seed):
    import or_cdchomp_amd
    rng = np.random.default_rng(5000 + seed)
    model, what = random_robot(seed)
    n_dof = model.n_dof
    # active dofs: all, or a subset (the others stay where the robot stands; their links' spheres may be inactive)
    if rng.uniform() < 0.35 and n_dof > 3:
        adofs = sorted(rng.choice(n_dof, size=int(rng.integers(2, n_dof)), replace=False).tolist())
    else:
        adofs = list(range(n_dof))
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
    dofvals = rng.uniform(0.5 * lo, 0.5 * hi)
    which = ("table", 2, 4, "table")[int(rng.integers(0, 4))]
    base = ([-0.55, 0.05, 0.75] if which == "table" else [0.05, -0.1, 0.35]) + list(_random_quat(rng, 0.7))
    floating = bool(rng.uniform() < 0.25)
    precision = 32 if rng.uniform() < 0.25 else 64
    momentum = bool(rng.uniform() < 0.3)
    second_order = bool(rng.uniform() < 0.12)
    hmc = momentum and bool(rng.uniform() < 0.4)
    long_traj = bool(rng.uniform() < 0.1)                 # several tiles of waypoints
    tol = 1e-3 if precision == 32 else 1e-6
    n_runs = (3, 3, 3, 40, 300)[int(rng.integers(0, 5))]           # 256 runs and more: the hmc streams live on the device
    n_points = int(rng.integers(100, 230)) if long_traj else int(rng.integers(5, 72))
    n_iter = int(rng.integers(6, 16))
    kw = dict(n_points=n_points, lambda_=float(rng.uniform(120.0, 400.0)), obs_factor=float(rng.uniform(20.0, 200.0)),
              obs_factor_self=float(rng.uniform(2.0, 20.0)), epsilon=float(rng.uniform(0.06, 0.14)),
              epsilon_self=float(rng.uniform(0.02, 0.08)))
    if momentum:
        kw["use_momentum"] = 1
    if second_order:
        kw["derivative"] = 2
    if hmc:
        kw["use_hmc"] = 1
        kw["hmc_resample_lambda"] = float(rng.uniform(0.02, 0.3))
    seeds = rng.integers(0, 1000, size=n_runs).astype(np.uint32)
    if floating:
        kw["floating_base"] = 1
    # now and then the batch is cut over several "devices" inside the process (contiguous blocks, host-side
    # gather; here the one card two or three times, uneven blocks included)
    shards = int(rng.integers(2, 4)) if rng.uniform() < 0.15 else 1
    mod = or_cdchomp_amd.Module([0] * shards if shards > 1 else 0)
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    grids, poses = _scene(mod, oracle, which)
    # the workgroup shapes a caller can ask for (orc_set_workgroup_threads, orc_set_workgroups_per_cu)
    threads = (0, 0, 192, 512)[int(rng.integers(0, 4))]
    per_cu = 4 if rng.uniform() < 0.3 and threads == 0 else 0
    mod.set_workgroup_threads(threads)
    if rng.uniform() < 0.2:
        mod.set_num_streams(2)                             # launches of one module on two streams: nothing a run can see
    mod.set_workgroups_per_cu(per_cu)
    goals = rng.uniform(0.7 * lo[adofs], 0.7 * hi[adofs], size=(n_runs, len(adofs)))
    basegoals = None
    if floating:
        basegoals = np.tile(np.asarray(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.2, 0.2, size=(n_runs, 3))
    # one call, or two (the run's iteration counter restarts, the next resampling iteration is kept: src/orcdchomp_mod.cpp:2752)
    calls = [n_iter] if rng.uniform() < 0.7 or n_iter < 4 else [n_iter // 2, n_iter - n_iter // 2]
    desc = "%s; %d runs, %d of %d dofs active, %s, fp%d, %s, %d iterations, %d threads, per_cu %d, %s" % (
        what, n_runs, len(adofs), n_dof, "floating" if floating else "fixed", precision, which, n_iter, threads, per_cu, dict(kw, calls=calls, shards=shards))
    rob = oracle.OraRobot(model)
    okw = dict(kw)
    if "derivative" in okw:
        okw["D"] = okw.pop("derivative")                  # the oracle's name for it
    # the reference refuses a robot whose active dofs move no sphere (src/orcdchomp_mod.cpp:2296): so must both sides
    try:
        probe = oracle.OraRun(rob, base, dofvals, adofs, goals[0], grids, poses, oracle.default_params(**okw),
                              basegoal=None if basegoals is None else basegoals[0])
    except RuntimeError as e:
        with pytest.raises(RuntimeError) as pe:
            mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, precision=precision, **kw)
        assert str(e) in str(pe.value)
        print("seed %d (%s): both refuse: %s" % (seed, what, e))
        return
    probe.destroy()
    # (a run the four-per-CU budget is not built for, or has no room for, keeps the default budget)
    bid = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, precision=precision, **kw)
    seeded = mod.batch_gettraj(bid)
    traces = []
    for n_call in calls:
        costs, status = mod.batch_iterate(bid, n_call)
        traces.append(mod.batch_trace(bid, n_call))
    trace = np.concatenate(traces, axis=1)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    errs = []
    for k in sorted(set([0, n_runs // 2, n_runs - 1, 1 % n_runs, (n_runs * 3) // 4])):      # (all of three, a sample of a larger batch)
        run = oracle.OraRun(rob, base, dofvals, adofs, goals[k], grids, poses, oracle.default_params(seed=int(seeds[k]), **okw),
                            basegoal=None if basegoals is None else basegoals[k])
        if precision == 64:
            assert np.array_equal(seeded[k], run.traj())
        otrace = []
        for n_call in calls:
            st, ocosts, otr = run.iterate(n_call, trace=True)
            otrace.append(otr)
            if st != 0:
                break
        otrace = np.concatenate(otrace, axis=0)
        assert st == status[k], (seed, desc, k, st, status[k])
        if st == 0:
            errs.append(common.rel_l2(traj[k], run.traj()))
            assert np.allclose(costs[k], ocosts, rtol=tol * (100 if precision == 32 else 1), atol=1e-12), (seed, desc, costs[k], ocosts)
            # the costs of every iteration (what `dat_filename` logs, src/orcdchomp_mod.cpp:2815-2818)
            assert np.allclose(trace[k], otrace, rtol=tol * (100 if precision == 32 else 1), atol=1e-12), (seed, desc, k)
        run.destroy()
    assert errs and max(errs) <= tol, (seed, desc, errs)
    print("seed %d (%s; %d runs, %d of %d dofs active, %s, fp%d, %d points, %s%s): worst rel L2 %.2e" % (
        seed, what, n_runs, len(adofs), n_dof, "floating" if floating else "fixed", precision, n_points,
        "%s fields" % which if which != "table" else "table", (", momentum" if momentum else "") + (" + hmc" if hmc else "") + (", derivative 2" if second_order else "")
        + (", %d threads" % threads if threads else "") + (", four per CU asked" if per_cu == 4 else ""), max(errs)))


@pytest.mark.parametrize("seed", SEEDS[:16] if len(SEEDS) <= 24 else SEEDS)
def test_random_robot_collision_verdict_matches_oracle(oracle, seed):
    """gettraj's re-check on the device (src/orcdchomp_mod.cpp:2958-3006): field contacts and
    self collisions of robots whose spheres overlap as they come -- which pairs of links count (same link, parent and
    child, links that touch at the zero configuration) is decided by two independent pieces of code."""
    import or_cdchomp_amd
    rng = np.random.default_rng(9000 + seed)
    model, what = random_robot(seed)
    n_dof = model.n_dof
    adofs = list(range(n_dof)) if rng.uniform() < 0.6 or n_dof < 4 else sorted(rng.choice(n_dof, size=int(rng.integers(2, n_dof)), replace=False).tolist())
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
    dofvals = rng.uniform(0.3 * lo, 0.3 * hi)
    which = ("table", 2)[int(rng.integers(0, 2))]
    base = ([-0.55, 0.05, 0.75] if which == "table" else [0.05, -0.1, 0.35]) + list(_random_quat(rng, 0.7))
    floating = bool(rng.uniform() < 0.25)
    n_runs, n_points = 12, int(rng.integers(6, 50))
    kw = dict(n_points=n_points, lambda_=200.0, obs_factor=50.0)
    if floating:
        kw["floating_base"] = 1
    mod = or_cdchomp_amd.Module(0)
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    vmax = rng.uniform(0.2, 3.0, size=n_dof)
    mod.set_velocity_limits(model.name, vmax)
    grids, poses = _scene(mod, oracle, which)
    goals = rng.uniform(lo[adofs], hi[adofs], size=(n_runs, len(adofs)))
    basegoals = None
    if floating:
        basegoals = np.tile(np.asarray(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.3, 0.3, size=(n_runs, 3))
    rob = oracle.OraRobot(model)
    try:
        oracle.OraRun(rob, base, dofvals, adofs, goals[0], grids, poses, oracle.default_params(**kw),
                      basegoal=None if basegoals is None else basegoals[0]).destroy()
    except RuntimeError:
        pytest.skip("the active dofs of this draw move no sphere")
    bid = mod.batch_create(model.name, goals, basegoals=basegoals, **kw)
    mod.batch_iterate(bid, 2)
    got = mod.batch_collision_verdict(bid)
    trajs = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    kinds = [0, 0, 0]
    for k in range(n_runs):
        orun = oracle.OraRun(rob, base, dofvals, adofs, goals[k], grids, poses, oracle.default_params(**kw),
                             basegoal=None if basegoals is None else basegoals[k])
        orun.set_traj(trajs[k])
        want = orun.collision_recheck(vmax[adofs])
        orun.destroy()
        mine = {q: got[q][k] for q in got}
        assert want["collides"] == mine["collides"], (seed, what, k, want, mine)
        if want["collides"]:
            assert want["sphere"] == mine["sphere"] and want["field"] == mine["field"], (seed, what, k, want, mine)
            assert np.isclose(want["time"], mine["time"], rtol=1e-12, atol=1e-15), (seed, what, k, want, mine)
            assert np.isclose(want["depth"], mine["depth"], rtol=1e-9, atol=1e-12), (seed, what, k, want, mine)
            kinds[1 if want["field"] >= 0 else 2] += 1
        else:
            kinds[0] += 1
    print("seed %d (%s, %s): %d free, %d in a field, %d self collisions" % (seed, what, "floating" if floating else "fixed", *kinds))


@pytest.mark.parametrize("seed", SEEDS[:16] if len(SEEDS) <= 24 else SEEDS)
def test_random_robot_tsr_constraint_matches_oracle(oracle, seed):
    """`con_tsr 'all link L'` (src/libcd/chomp.c:553-600, src/orcdchomp_mod.cpp:1330-1497) on random
    robots: one to three rows of the pose error of the last link held at their start values on every moving point.  The
    constraint step's elimination runs in registers in one of three shapes chosen from the run's size (csrc/tsr.h)."""
    import or_cdchomp_amd
    rng = np.random.default_rng(13000 + seed)
    model, what = random_robot(seed)
    n_dof = model.n_dof
    adofs = list(range(n_dof))
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
    dofvals = rng.uniform(0.5 * lo, 0.5 * hi)
    # a unit quaternion to the last bit matters to nobody, but the poses are inverted as if it were one
    base = [-0.55, 0.05, 0.75] + list(_random_quat(rng, 0.7))
    link = model.link_names[-1]
    li = len(model.link_names) - 1
    # joints between the base and that link
    n_anc, cur = 0, li
    while cur >= 0:
        n_anc += model.joint_type[cur] != robots.JOINT_FIXED
        cur = model.parent[cur]
    k = int(rng.integers(1, 1 + min(3, max(1, n_anc // 2))))
    floating = bool(rng.uniform() < 0.2)
    # the constraint on every moving point, the start point itself a variable held on a TSR (`start_tsr`,
    # src/orcdchomp_mod.cpp:1988-1992, 2316-2323, 2570-2576: not together with a floating base), or both
    variant = "con" if floating else ("con", "con", "con", "start", "start+con")[int(rng.integers(0, 5))]
    momentum = bool(rng.uniform() < 0.3)
    n_runs = 3
    n_points = int(rng.integers(100, 210)) if rng.uniform() < 0.15 else int(rng.integers(5, 70))
    n_iter = int(rng.integers(5, 14))
    lam = float(rng.uniform(120.0, 400.0))
    mod = or_cdchomp_amd.Module(0)
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    grids, poses = _scene(mod, oracle, "table")
    rob = oracle.OraRobot(model)
    R, t, _, _ = rob.fk(base, dofvals)
    goals = dofvals[None, :] + 0.35 * rng.uniform(-1, 1, size=(n_runs, n_dof)) * np.minimum(1.0, hi - lo)
    goals = np.ascontiguousarray(np.clip(goals, lo, hi))
    # rows the link's joints can actually move: a row whose Jacobian is rounding noise (the y of a point on its own
    # joint's axis) makes the step a quotient of two such numbers, in the reference as here
    def bw_of(rows):
        return [[0, 0] if r in rows else ([-1, 1] if r < 3 else [-3, 3]) for r in range(6)]

    def draw_rows(points, count, among=range(6), together_with=()):
        """`count` rows out of `among` whose Jacobian, stacked on that of the rows `together_with`, has full rank at `points`"""
        for attempt in range(12):
            rows = sorted(rng.choice(list(among), size=count, replace=False).tolist())
            probe = oracle.OraRun(rob, base, dofvals, adofs, goals[0], grids, poses, oracle.default_params(n_points=9))
            probe.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], oracle.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], bw_of(sorted(set(rows) | set(together_with))))
            smin = min(np.linalg.svd(probe.eval_contsr(0, probe.traj()[i])[1], compute_uv=False).min() for i in points)
            probe.destroy()
            if smin > 0.02:
                return rows, bw_of(rows)
        pytest.skip("no well-posed rows found for the last link of this draw")
    rows, Bw = draw_rows(range(1, 8), k)
    rows_s, Bw_s = None, None
    if variant == "start":
        rows_s, Bw_s = draw_rows([0], k)
    elif variant == "start+con":
        # the start point carries both constraints: other rows than the constraint of every point holds, or the reference's
        # system has the same row twice (singular to rounding: dgesv's answer is noise there, and so is everybody's)
        if 2 * k > n_anc:
            k = max(1, n_anc // 2); rows, Bw = draw_rows(range(1, 8), k)
        free = [r for r in range(6) if r not in rows]
        ks = int(rng.integers(1, 1 + max(1, min(len(free), n_anc - k, 3))))
        rows_s, Bw_s = draw_rows([0], ks, among=free, together_with=rows)
    tsr = robots.Tsr(T0w_R=R[li], T0w_d=t[li], Bw=Bw)
    cmd = "createbatch robot %s n_runs %d adofgoals 0x%x n_points %d lambda %.17g obs_factor 100" % (
        model.name, n_runs, goals.ctypes.data, n_points, lam)
    if variant != "start":
        cmd += " con_tsr 'all link %s' '%s'" % (link, tsr.serialize())
    if variant != "con":
        # `start_tsr` speaks of the active manipulator's end effector: the last link, no tool offset
        mod.add_manipulator(model.name, "tip", li)
        cmd += " start_tsr '%s'" % robots.Tsr(T0w_R=R[li], T0w_d=t[li], Bw=Bw_s).serialize()
    basegoals = None
    if floating:
        basegoals = np.tile(np.asarray(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.15, 0.15, size=(n_runs, 3))
        cmd += " basegoals 0x%x floating_base" % basegoals.ctypes.data
    if momentum:
        cmd += " use_momentum"
    bid = int(mod.SendCommand(cmd))
    seeded = mod.batch_gettraj(bid)
    costs, status = mod.batch_iterate(bid, n_iter)
    ptrace = mod.batch_trace(bid, n_iter)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    okw = dict(n_points=n_points, lambda_=lam, obs_factor=100.0)
    if floating:
        okw["floating_base"] = 1
    if momentum:
        okw["use_momentum"] = 1
    errs = []
    for r in range(n_runs):
        st_arg = None if variant == "con" else (li, [0, 0, 0, 0, 0, 0, 1], oracle.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], Bw_s)
        run = oracle.OraRun(rob, base, dofvals, adofs, goals[r], grids, poses, oracle.default_params(start_tsr=st_arg, **okw),
                            basegoal=None if basegoals is None else basegoals[r])
        if variant != "start":
            assert run.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], oracle.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], Bw) == k
        assert np.array_equal(seeded[r], run.traj()), (seed, what, variant)
        st, oc, otr = run.iterate(n_iter, trace=True)
        ot = run.traj().copy()
        run.destroy()
        # where the two part company, if they do: at once is a mistake, late and growing is the run's own conditioning
        rel = np.abs(ptrace[r] - otr).max(axis=1) / np.maximum(np.abs(otr).max(axis=1), 1e-300)
        part = "first iteration with costs apart by 1e-9: %s of %d; rel per iteration %s" % (
            (np.where(rel > 1e-9)[0][:1].tolist() or ["none"])[0], n_iter, np.array2string(rel, precision=1))
        if st != 0 or not np.all(np.isfinite(ot)):
            assert status[r] != 0 or not np.all(np.isfinite(traj[r])), (seed, what, variant, r, st, status[r], part)
            continue
        assert status[r] == 0, (seed, what, variant, r, status[r], part)
        err = common.rel_l2(traj[r], ot)
        allowed = 1e-6
        if err > 1e-8:
            # a run that parts company late and growing: how far does the ORACLE move when its goal moves by one ulp?  (the rule of
            # every other parity test, tests/common.py: a constrained run whose system turns nearly singular on the way amplifies
            # rounding like a run that bounces off its limits; draw 297 of the round-6 wide run: 1.006e-6)
            amp = 0.0
            for f in common.ULPS:
                run2 = oracle.OraRun(rob, base, dofvals, adofs, goals[r] * f, grids, poses, oracle.default_params(start_tsr=st_arg, **okw),
                                     basegoal=None if basegoals is None else basegoals[r])
                if variant != "start":
                    run2.add_contsr(li, [0, 0, 0, 0, 0, 0, 1], oracle.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], Bw)
                st2, _ = run2.iterate(n_iter)
                if st2 == 0:
                    amp = max(amp, common.rel_l2(run2.traj(), ot))
                run2.destroy()
            allowed = max(1e-6, common.CHAOS_FACTOR * amp)
            print("seed %d run %d: rel L2 %.2e, the oracle's own amplification under one-ulp changes of the goal %.2e" % (seed, r, err, amp))
        assert err <= allowed, (seed, what, rows, r, err, allowed, part)
        errs.append(err)
        if err <= 1e-8:
            assert np.allclose(costs[r], oc, rtol=1e-6, atol=1e-12), (seed, what, variant, rows, rows_s, r, costs[r], oc, part)
    if not errs:
        pytest.skip("the constraint of this draw is singular for the oracle as well")
    print("seed %d (%s, %s%s, %d points, %s, rows %s%s of link %s behind %d joints): worst rel L2 %.2e" % (
        seed, what, "floating" if floating else "fixed", ", momentum" if momentum else "", n_points, variant, rows if variant != "start" else "",
        " start rows %s" % rows_s if rows_s else "", link, n_anc, max(errs)))


@pytest.mark.parametrize("seed", SEEDS[:12] if len(SEEDS) <= 24 else SEEDS)
def test_random_robot_through_the_commands(oracle, seed):
    """the command layer on random robots: `create` / `iterate` / `gettraj` of one run give the
    bits of the same run inside a batch; the trajectory document names the active dofs and carries the retimed waypoints
    (src/orcdchomp_mod.cpp:2897-2956) digit for digit; fed back through `create starttraj` (src/orcdchomp_mod.cpp:2375-2416)
    it seeds what the oracle's sampler reads out of the same document, at this and at another length."""
    import re
    import or_cdchomp_amd
    from or_cdchomp_amd import bindings
    rng = np.random.default_rng(17000 + seed)
    model, what = random_robot(seed)
    n_dof = model.n_dof
    adofs = list(range(n_dof)) if rng.uniform() < 0.5 or n_dof < 4 else sorted(rng.choice(n_dof, size=int(rng.integers(2, n_dof)), replace=False).tolist())
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
    dofvals = rng.uniform(0.5 * lo, 0.5 * hi)
    base = [-0.55, 0.05, 0.75] + list(_random_quat(rng, 0.7))
    floating = bool(rng.uniform() < 0.3)
    momentum = bool(rng.uniform() < 0.3)
    n_points, n_iter = int(rng.integers(4, 60)), int(rng.integers(1, 9))
    lam = round(float(rng.uniform(120.0, 400.0)), 4)           # the python layer writes `lambda %0.04f`
    mod = bindings.bind(or_cdchomp_amd.Module(0))
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    vmax = rng.uniform(0.2, 3.0, size=n_dof)
    mod.set_velocity_limits(model.name, vmax)
    _scene(mod, oracle, "table")
    n_runs, mine = 5, int(rng.integers(0, 5))
    goals = rng.uniform(0.7 * lo[adofs], 0.7 * hi[adofs], size=(n_runs, len(adofs)))
    basegoals = None
    if floating:
        basegoals = np.tile(np.asarray(base), (n_runs, 1)); basegoals[:, :3] += rng.uniform(-0.2, 0.2, size=(n_runs, 3))
    kw = dict(n_points=n_points, lambda_=lam, obs_factor=80.0)
    ckw = dict(kw)
    if momentum:
        kw["use_momentum"] = 1; ckw["use_momentum"] = True
    if floating:
        kw["floating_base"] = 1
    try:
        bid = mod.batch_create(model.name, goals, basegoals=basegoals, **kw)
    except RuntimeError as e:
        assert "at least one sphere" in str(e)
        with pytest.raises(RuntimeError, match="at least one sphere"):
            mod.create(robot=model.name, adofgoal=[float(v) for v in goals[mine]], **ckw)
        return
    bcosts, bstatus = mod.batch_iterate(bid, n_iter)
    btraj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    if floating:
        run = mod.create(robot=model.name, adofgoal=[float(v) for v in goals[mine]], basegoal=[float(v) for v in basegoals[mine]], floating_base=True, **ckw)
    else:
        run = mod.create(robot=model.name, adofgoal=[float(v) for v in goals[mine]], **ckw)
    if bstatus[mine] != 0:
        with pytest.raises(RuntimeError, match="Resulting trajectory is outside of joint limits!"):
            mod.iterate(run=run, n_iter=n_iter)
        mod.destroy(run=run)
        return
    out = [None]
    mod.iterate(run=run, n_iter=n_iter, cost=out)                  # the python layer's out-parameter (orcdchomp.py:176-185)
    cost = out[0]
    traj = mod.batch_gettraj(int(run))[0]
    assert np.array_equal(traj, btraj[mine])
    assert np.isclose(cost, bcosts[mine][0], rtol=1e-5, atol=0)      # operator<<(double): six digits
    text = mod.gettraj(run=run, no_collision_check=True)
    count = int(re.search(r'<data count="(\d+)">', text).group(1))
    vals = np.array(re.search(r'<data count="\d+">\s*(.*?)\s*</data>', text, re.S).group(1).split(), dtype=float).reshape(count, -1)
    groups = {m.group(1).split()[0]: (int(m.group(2)), int(m.group(3)), m.group(1))
              for m in re.finditer(r'<group name="([^"]+)" offset="(\d+)" dof="(\d+)"', text)}
    na, c0 = len(adofs), (7 if floating else 0)
    assert count == n_points and groups["joint_values"][:2] == (0, na) and groups["deltatime"][:2] == (na, 1)
    assert groups["joint_values"][2] == "joint_values %s %s" % (model.name, " ".join(str(d) for d in adofs))
    assert np.array_equal(vals[:, :na], traj[:, c0:])
    dt = vals[:, na]
    want_dt = np.r_[0.0, (np.abs(np.diff(traj[:, c0:], axis=0)) / vmax[adofs]).max(axis=1)]
    assert np.allclose(dt, want_dt, rtol=1e-12, atol=0)
    if floating:
        assert vals.shape[1] == na + 1 + 14 and groups["affine_transform"][:2] == (na + 1, 7)
        want = oracle.gettraj_affine_groups(traj, dt)
        assert np.array_equal(vals[:, na + 1:na + 8], want[:, 1:8]) and np.allclose(vals[:, na + 8:], want[:, 8:15], rtol=1e-15, atol=0)
    else:
        assert vals.shape[1] == na + 1
    # and back in: the same length gives the same waypoints, another length the oracle's samples
    for npts in (n_points, int(rng.integers(3, 80))):
        if dt.sum() == 0.0:
            break
        run2 = mod.create(robot=model.name, starttraj=text, n_points=npts, **({"floating_base": True} if floating else {}))
        t2 = mod.batch_gettraj(int(run2))[0]
        mod.destroy(run=run2)
        if floating:
            want2 = oracle.sample_starttraj_floating(vals[:, :na], vals[:, na + 1:na + 8], dt, npts)
        else:
            want2 = oracle.sample_starttraj(vals[:, :na], dt, npts)
        assert t2.shape == want2.shape
        assert np.allclose(t2, want2, rtol=1e-13, atol=1e-14), (seed, what, npts, np.abs(t2 - want2).max())
        # (sampled at equal steps of time, not at the document's waypoints: the ends are the ends)
        assert np.allclose(t2[0], traj[0], rtol=1e-12, atol=1e-14) and np.allclose(t2[-1], traj[-1], rtol=1e-12, atol=1e-14)
    mod.destroy(run=run)
    print("seed %d (%s; %d of %d dofs, %s, %d points): commands ok" % (seed, what, na, n_dof, "floating" if floating else "fixed", n_points))
