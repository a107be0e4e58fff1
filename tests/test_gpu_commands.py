"""-m gpu: the setup-side commands and edge cases of the path through SendCommand:
cache file format, addfield_fromobsarray, removefield, tiny trajectories, robots outside every field."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from or_cdchomp_amd import bindings, robots

pytestmark = pytest.mark.gpu


def _mk():
    import or_cdchomp_amd
    return or_cdchomp_amd.Module(0)


def test_cache_file_round_trip(oracle, tmp_path):
    """raw doubles, C order, no header, validated by size only (reference src/orcdchomp_mod.cpp:416-444,571-580)"""
    from or_cdchomp_amd import scenes
    mod2 = bindings.bind(_mk())
    scenes.add_tabletop(mod2)
    cache = str(tmp_path / "sdf_tablemug.dat")
    mod2.computedistancefield(kinbody="table", cache_filename=cache)
    data, lengths, pose = mod2.get_sdf("table")
    raw = np.fromfile(cache, dtype=np.float64)
    assert raw.size == data.size and os.path.getsize(cache) == data.size * 8
    assert np.array_equal(raw.reshape(data.shape), data)
    prob = common.tabletop_problem(oracle)
    assert np.array_equal(data, prob["sdf"].data)
    # a second module reads the cache instead of recomputing: poison one cell to prove it
    raw2 = raw.copy(); raw2[5] = 123.456
    raw2.tofile(cache)
    mod3 = bindings.bind(_mk())
    scenes.add_tabletop(mod3)
    mod3.computedistancefield(kinbody="table", cache_filename=cache, require_cache=True)
    d3, _, _ = mod3.get_sdf("table")
    assert d3.reshape(-1)[5] == 123.456
    # wrong size -> recomputed; with require_cache -> the reference's exception
    raw[:100].tofile(cache)
    mod4 = bindings.bind(_mk())
    scenes.add_tabletop(mod4)
    with pytest.raises(RuntimeError, match="Field not found from cache, but require_cache flag set!"):
        mod4.computedistancefield(kinbody="table", cache_filename=cache, require_cache=True)
    mod4.computedistancefield(kinbody="table", cache_filename=cache)
    d4, _, _ = mod4.get_sdf("table")
    assert np.array_equal(d4, data)


def test_addfield_fromobsarray_and_removefield(oracle):
    mod = bindings.bind(_mk())
    model, base, dofvals, adofs = common.wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    from or_cdchomp_amd import scenes
    mod.add_kinbody_boxes("blob", [(scenes.IDENT, [0.1, 0.1, 0.1])], transform=[0.0, 0.0, 0.6, 0, 0, 0, 1])
    occ = np.zeros((20, 20, 20)); occ[8:12, 8:12, 8:12] = np.inf
    occ = np.ascontiguousarray(occ)
    lengths = [0.8, 0.8, 0.8]
    pose = [-0.4, -0.4, -0.4, 0, 0, 0, 2.0]          # quaternion gets normalised (mod.cpp:682)
    mod.addfield_fromobsarray(kinbody="blob", obsarray="%#x" % occ.ctypes.data, sizes=occ.shape, lengths=lengths, pose=pose)
    data, ln, ps = mod.get_sdf("blob")
    assert np.array_equal(data, oracle.OraGrid(occ, lengths).bin_sdf().data)
    assert np.allclose(ps, [-0.4, -0.4, -0.4, 0, 0, 0, 1.0])
    with pytest.raises(RuntimeError, match="We already have an sdf for this kinbody!"):
        mod.addfield_fromobsarray(kinbody="blob", obsarray="%#x" % occ.ctypes.data, sizes=occ.shape, lengths=lengths)
    goal = common.wam_goals(1, seed=4)[0]
    run = mod.create(robot=model.name, adofgoal=list(goal), n_points=20)
    cost = [None]
    mod.iterate(run=run, n_iter=5, cost=cost)
    assert np.isfinite(cost[0])
    mod.destroy(run=run)
    mod.removefield(kinbody="blob")
    with pytest.raises(RuntimeError, match="No signed distance fields have yet been computed!"):
        mod.create(robot=model.name, adofgoal=list(goal))
    with pytest.raises(RuntimeError, match="you must pass a created run!"):
        mod.destroy(run=run)


@pytest.mark.parametrize("n_points", [3, 4, 7, 65, 130, 259, 300])
def test_trajectory_lengths(oracle, n_points):
    """m = 1 (a single moving waypoint) up to more waypoints than one tile holds, and past the 256
    moving waypoints the scan solve and the one-wavefront limit rounds cover (cyclic reduction and the
    general limit loop take over)"""
    mod = _mk()
    model = common.setup_product_wam(mod)
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    goals = common.wam_goals(3, seed=n_points)
    kw = dict(n_points=n_points, lambda_=100.0, obs_factor=500.0)
    bid = mod.batch_create(model.name, goals, **kw)
    costs, status = mod.batch_iterate(bid, 12)
    traj = mod.batch_gettraj(bid)
    otraj, ocosts, ostatus, _ = oracle.batch_run(oracle.OraRobot(model), base, dofvals, adofs, goals, [prob["sdf"]],
                                                 [prob["pose"]], oracle.default_params(**kw), 12)
    assert np.array_equal(status, ostatus)
    for k in range(3):
        assert common.rel_l2(traj[k], otraj[k]) <= 1e-6, (k, common.rel_l2(traj[k], otraj[k]))
    assert np.allclose(costs, ocosts, rtol=1e-6, atol=0)
    with pytest.raises(RuntimeError, match="n_points must be >=3!"):
        mod.batch_create(model.name, goals, n_points=2)


def test_robot_outside_every_field(oracle):
    """spheres outside the field contribute no obstacle term (reference src/orcdchomp_mod.cpp:1180-1182);
    also the demo's own quirk: a field computed for a body without geometry is a 10^3 cube of free space"""
    mod = _mk()
    model, base, dofvals, adofs = common.wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    mod.SendCommand("computedistancefield kinbody %s" % model.name)     # robots carry no box geometry here
    data, lengths, pose = mod.get_sdf(model.name)
    assert data.shape == (10, 10, 10) and np.allclose(lengths, 0.4)
    assert np.isinf(data).all()                                          # no obstacle anywhere: +inf everywhere
    goals = common.wam_goals(2, seed=8)
    kw = dict(n_points=30, lambda_=100.0)
    bid = mod.batch_create(model.name, goals, **kw)
    costs, status = mod.batch_iterate(bid, 10)
    traj = mod.batch_gettraj(bid)
    grid = oracle.OraGrid(data, lengths)
    pw = np.zeros(7)
    oracle.lib().ora_kin_pose_compose(oracle.dp(oracle.f64(base)), oracle.dp(oracle.f64(pose)), oracle.dp(pw))
    otraj, ocosts, ostatus, _ = oracle.batch_run(oracle.OraRobot(model), base, dofvals, adofs, goals, [grid], [pw],
                                                 oracle.default_params(**kw), 10)
    assert np.array_equal(status, ostatus)
    assert max(common.rel_l2(traj[k], otraj[k]) for k in range(2)) <= 1e-6
    assert np.allclose(costs, ocosts, rtol=1e-6, atol=0)


def test_device_sdf_build_bit_exact(oracle, monkeypatch):
    """the GPU distance-transform path produces the same bits as the host path and
    as the oracle's bin_sdf; also timed against the host on a 1-cm-cell sized grid"""
    import time
    from or_cdchomp_amd import scenes
    monkeypatch.setenv("ORC_SDF_DEVICE", "1")
    mod = bindings.bind(_mk())
    scenes.add_tabletop(mod)
    mod.computedistancefield(kinbody="table")
    data, _, _ = mod.get_sdf("table")
    assert np.array_equal(data, common.tabletop_problem(oracle)["sdf"].data)
    # a larger random occupancy through addfield_fromobsarray, device vs host vs oracle
    rng = np.random.default_rng(42)
    shape = (96, 80, 72)
    occ = np.ascontiguousarray(np.where(rng.uniform(size=shape) < 0.002, np.inf, 0.0))
    occ[10:30, 20:40, 5:25] = np.inf
    lengths = [0.96, 0.8, 0.72]
    mod.add_kinbody_boxes("big", [(scenes.IDENT, [0.1, 0.1, 0.1])], transform=scenes.IDENT)
    t0 = time.perf_counter()
    mod.addfield_fromobsarray(kinbody="big", obsarray="%#x" % occ.ctypes.data, sizes=shape, lengths=lengths)
    t_dev = time.perf_counter() - t0
    dev, _, _ = mod.get_sdf("big")
    monkeypatch.setenv("ORC_SDF_DEVICE", "0")
    mod.add_kinbody_boxes("big2", [(scenes.IDENT, [0.1, 0.1, 0.1])], transform=scenes.IDENT)
    t0 = time.perf_counter()
    mod.addfield_fromobsarray(kinbody="big2", obsarray="%#x" % occ.ctypes.data, sizes=shape, lengths=lengths)
    t_host = time.perf_counter() - t0
    host, _, _ = mod.get_sdf("big2")
    assert np.array_equal(dev, host)
    assert np.array_equal(dev, oracle.OraGrid(occ, lengths).bin_sdf().data)
    print("sdf build %s: device %.1f ms, host %.1f ms" % (shape, 1e3 * t_dev, 1e3 * t_host))


def _doc(waypoints, deltatimes):
    rows = " ".join(" ".join(repr(float(v)) for v in wp) + " " + repr(float(dt)) for wp, dt in zip(waypoints, deltatimes))
    return '<trajectory>\n<data count="%d">\n%s\n</data>\n</trajectory>\n' % (len(waypoints), rows)


def test_gettraj_retime_and_collision_verdict(oracle):
    """gettraj: linear retiming at the dof velocity limits and the sphere-vs-field verdict
    (reference src/orcdchomp_mod.cpp:2905-3006)"""
    import re
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    vmax = np.ones(model.n_dof); vmax[:7] = [0.5, 1.0, 2.0, 1.0, 4.0, 1.0, 0.25]
    mod.set_velocity_limits(model.name, vmax)
    goal = [0.6, -1.2, 0.3, 1.6, -0.4, 0.5, 0.2]
    run = mod.create(robot=model.name, adofgoal=goal, n_points=30, lambda_=100.0, obs_factor=500.0)
    mod.iterate(run=run, n_iter=10)
    text = mod.gettraj(run=run, no_collision_check=True)
    vals = np.array(re.search(r'<data count="30">\s*(.*?)\s*</data>', text, re.S).group(1).split(), dtype=float).reshape(30, 8)
    wp, dt = vals[:, :7], vals[:, 7]
    want = np.r_[0.0, (np.abs(np.diff(wp, axis=0)) / vmax[:7]).max(axis=1)]
    assert np.allclose(dt, want, rtol=1e-12, atol=0)
    mod.destroy(run=run)
    # a goal that drives the forearm through the table top: the verdict must be "in collision"
    model_, base, dofvals, adofs = common.wam_state()
    deep = [1.2, -0.2, 0.0, 0.3, 0.0, 0.0, 0.0]
    run = mod.create(robot=model.name, adofgoal=deep, n_points=30, lambda_=100.0, obs_factor=0.0, obs_factor_self=0.0)
    with pytest.raises(RuntimeError, match="Resulting trajectory is in collision!"):
        mod.gettraj(run=run)
    text = mod.gettraj(run=run, no_collision_exception=True)
    assert "Collision at t=" in mod.last_collision_details() and "table" in mod.last_collision_details()
    mod.gettraj(run=run, no_collision_exception=True, no_collision_details=True)
    assert mod.last_collision_details() == ""
    mod.destroy(run=run)


def test_batched_collision_verdict_matches_gettraj(oracle):
    """the device verdict of a whole batch (orc_batch_collision_verdict, `gettrajbatch ... verdict %p`)
    against the host re-check of `gettraj`, run by run: same first contact (time, sphere, field, depth)"""
    import re
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    vmax = np.ones(model.n_dof); vmax[:7] = [0.5, 1.0, 2.0, 1.0, 4.0, 1.0, 0.25]
    mod.set_velocity_limits(model.name, vmax)
    n_runs = 48
    goals = common.wam_goals(n_runs, seed=99)
    goals[:8, 1] = np.linspace(-0.4, 0.6, 8)          # a few that sweep the forearm through the table
    goals[:8, 0] = 1.2; goals[:8, 3] = 0.3
    kw = dict(n_points=40, lambda_=100.0, obs_factor=20.0)
    bid = mod.batch_create(model.name, goals, **kw)
    mod.batch_iterate(bid, 3)
    got = mod.batch_collision_verdict(bid)
    # the command form: waypoints and verdicts in one call
    out = np.zeros((n_runs, 40, 7)); ver = np.full(n_runs, -7, dtype=np.int32)
    mod.SendCommand("gettrajbatch run %d out 0x%x verdict 0x%x" % (bid, out.ctypes.data, ver.ctypes.data))
    assert np.array_equal(ver, got["collides"])
    assert np.array_equal(out, mod.batch_gettraj(bid))
    mod.batch_destroy(bid)
    assert 0 < got["collides"].sum() < n_runs, got["collides"]
    # against the oracle's restatement of the re-check loop (oracle/ora_run.c: ora_run_collision_recheck,
    # reference src/orcdchomp_mod.cpp:2958-3006) walking the very trajectories the device walked
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    rob = oracle.OraRobot(model)
    for k in range(n_runs):
        orun = oracle.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]], oracle.default_params(**kw))
        orun.set_traj(out[k])
        want = orun.collision_recheck(vmax[:7])
        orun.destroy()
        assert want["collides"] == got["collides"][k], (k, want)
        if want["collides"]:
            assert want["sphere"] == got["sphere"][k] and want["field"] == got["field"][k], (k, want)
            assert np.isclose(want["time"], got["time"][k], rtol=1e-12, atol=1e-15), (k, want, got["time"][k])
            assert np.isclose(want["depth"], got["depth"][k], rtol=1e-9, atol=1e-13), (k, want, got["depth"][k])
    for k in range(n_runs):
        run = mod.create(robot=model.name, adofgoal=list(goals[k]), **kw)
        mod.iterate(run=run, n_iter=3)
        mod.gettraj(run=run, no_collision_exception=True)
        details = mod.last_collision_details()
        mod.destroy(run=run)
        if not got["collides"][k]:
            assert details == "", (k, details)
            continue
        m = re.match(r"Collision at t=(\S+): sphere (\d+) of \S+ is (\S+) m inside the field of (\S+)", details)
        assert m, (k, details)
        assert int(m.group(2)) == got["sphere"][k] and m.group(4) == "table" and got["field"][k] == 0
        assert np.isclose(float(m.group(1)), got["time"][k], rtol=1e-5, atol=1e-9), (k, details, got["time"][k])
        assert np.isclose(float(m.group(3)), got["depth"][k], rtol=1e-4, atol=1e-9), (k, details, got["depth"][k])


def test_starttraj_seeding(oracle):
    """create starttraj: the run starts from the passed trajectory sampled at i*duration/(n_points-1)
    (reference src/orcdchomp_mod.cpp:2375-2416)"""
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    goal = common.wam_goals(1, seed=21)[0]
    kw = dict(n_points=40, lambda_=100.0, obs_factor=500.0)
    a = mod.batch_create(model.name, goal, **kw)
    mod.batch_iterate(a, 15)
    ta = mod.batch_gettraj(a)[0]
    # untimed document (deltatimes 0): sampled uniformly over the waypoint index -> identical waypoints
    run = mod.create(robot=model.name, starttraj=_doc(ta, np.zeros(40)), **kw)
    tb = mod.batch_gettraj(int(run))[0]
    assert np.array_equal(tb, ta)
    ca, _ = mod.batch_iterate(a, 5)
    cb, _ = mod.batch_iterate(int(run), 5)
    # (`create` runs its one run on the latency shape: the same trajectory bit for bit, cost sums grouped differently)
    assert np.array_equal(mod.batch_gettraj(a)[0], mod.batch_gettraj(int(run))[0]) and np.allclose(ca, cb, rtol=1e-13, atol=0)
    # timed document with non-uniform deltatimes, resampled to a different n_points
    wp = np.array([[0.0] * 7, [1.0] * 7, [3.0] * 7])
    run2 = mod.create(robot=model.name, starttraj=_doc(wp, [0.0, 1.0, 1.0]), n_points=5)
    t2 = mod.batch_gettraj(int(run2))[0]
    assert np.allclose(t2[:, 0], [0.0, 0.5, 1.0, 2.0, 3.0])
    # against the oracle's restatement of the sampling (oracle/ora_run.c: ora_sample_starttraj, reference
    # src/orcdchomp_mod.cpp:2375-2416): random waypoints inside the limits, irregular deltatimes (one zero)
    rng = np.random.default_rng(23)
    lo = np.asarray(model.limit_lower[:7]) + 0.2; hi = np.asarray(model.limit_upper[:7]) - 0.2
    wp3 = rng.uniform(lo, hi, size=(9, 7))
    dt3 = np.r_[0.0, rng.uniform(0.05, 0.9, size=8)]; dt3[4] = 0.0
    for npts in (5, 33, 101):
        run3 = mod.create(robot=model.name, starttraj=_doc(wp3, dt3), n_points=npts)
        t3 = mod.batch_gettraj(int(run3))[0]
        mod.destroy(run=run3)
        assert np.allclose(t3, oracle.sample_starttraj(wp3, dt3, npts), rtol=1e-14, atol=1e-15)
    with pytest.raises(RuntimeError, match="Cannot pass both adofgoal and starttraj!"):
        mod.SendCommand("create robot %s adofgoal '0 0 0 0 0 0 0' starttraj 'x'" % model.name)
    # orc_batch_set_traj: warm start of a whole batch
    goals = common.wam_goals(3, seed=22)
    b1 = mod.batch_create(model.name, goals, **kw); mod.batch_iterate(b1, 8)
    b2 = mod.batch_create(model.name, goals, **kw); mod.batch_set_traj(b2, mod.batch_gettraj(b1))
    c1, _ = mod.batch_iterate(b1, 4); c2, _ = mod.batch_iterate(b2, 4)
    assert np.array_equal(c1, c2)


def _groups(text):
    """(rows [count][width], {group name's first word: (offset, dof)}) of a trajectory document"""
    import re
    count = int(re.search(r'<data count="(\d+)">', text).group(1))
    vals = np.array(re.search(r'<data count="\d+">\s*(.*?)\s*</data>', text, re.S).group(1).split(), dtype=float)
    groups = {m.group(1).split()[0]: (int(m.group(2)), int(m.group(3)), m.group(1))
              for m in re.finditer(r'<group name="([^"]+)" offset="(\d+)" dof="(\d+)"', text)}
    return vals.reshape(count, -1), groups


def test_floating_base_wire_formats(oracle):
    """gettraj of a floating-base run carries the base pose as `affine_transform` / `affine_velocities` groups
    (reference src/orcdchomp_mod.cpp:2912-2956: x y z then the quaternion w x y z, velocities = differences over the
    waypoint's deltatime), and `create starttraj ... floating_base` samples both groups (src/orcdchomp_mod.cpp:2378-2404);
    both against the oracle's restatements (oracle/ora_run.c)"""
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    _, base, _, _ = common.wam_state()
    goal = common.wam_goals(1, seed=31)[0]
    basegoal = np.array(base, dtype=float); basegoal[:3] += [0.2, -0.1, 0.15]
    kw = dict(n_points=30, lambda_=100.0, obs_factor=500.0)
    run = mod.create(robot=model.name, adofgoal=list(goal), basegoal=list(basegoal), floating_base=True, **kw)
    mod.iterate(run=run, n_iter=12)
    traj = mod.batch_gettraj(int(run))[0]                       # [30][14], base in libcd's order x y z qx qy qz qw
    text = mod.gettraj(run=run, no_collision_check=True)
    rows, groups = _groups(text)
    assert rows.shape == (30, 7 + 1 + 14)
    assert groups["joint_values"][:2] == (0, 7) and groups["deltatime"][:2] == (7, 1)
    assert groups["affine_transform"][:2] == (8, 7) and groups["affine_velocities"][:2] == (15, 7)
    assert groups["affine_transform"][2] == "affine_transform %s 39" % model.name       # DOF_XYZ | DOF_RotationQuat
    assert np.array_equal(rows[:, :7], traj[:, 7:])                                         # the arm columns, digit for digit
    want = oracle.gettraj_affine_groups(traj, rows[:, 7])
    assert np.array_equal(rows[:, 7], want[:, 0]) and np.array_equal(rows[:, 8:15], want[:, 1:8])
    assert np.allclose(rows[:, 15:22], want[:, 8:15], rtol=1e-15, atol=0)
    assert np.array_equal(rows[:, 8:11], traj[:, 0:3]) and np.array_equal(rows[:, 11], traj[:, 6])      # x y z, then qw first
    assert np.all(rows[0, 15:22] == 0.0)

    # back in through starttraj: resampled at i * duration / (n_points - 1) like the reference
    for npts in (30, 17, 64):
        run2 = mod.create(robot=model.name, starttraj=text, floating_base=True, n_points=npts)
        t2 = mod.batch_gettraj(int(run2))[0]
        want2 = oracle.sample_starttraj_floating(rows[:, :7], rows[:, 8:15], rows[:, 7], npts)
        assert t2.shape == (npts, 14)
        assert np.allclose(t2, want2, rtol=1e-14, atol=1e-15), np.abs(t2 - want2).max()
        assert np.allclose(np.linalg.norm(t2[:, 3:7], axis=1), 1.0, rtol=1e-15)
        assert np.allclose(t2[0], traj[0], rtol=1e-14, atol=1e-15) and np.allclose(t2[-1], traj[-1], rtol=1e-14, atol=1e-15)
        # the run is usable: it iterates from there
        mod.iterate(run=run2, n_iter=3)
        mod.destroy(run=run2)
    # a document without the base group is refused
    with pytest.raises(RuntimeError, match="affine_transform"):
        mod.create(robot=model.name, starttraj=_doc(rows[:, :7], rows[:, 7]), floating_base=True, n_points=30)
    mod.destroy(run=run)


def test_self_collision_in_the_recheck(oracle):
    """gettraj's re-check also reports two spheres on links that may collide overlapping (reference
    `|| boostrobot->CheckSelfCollision(report)`, src/orcdchomp_mod.cpp:2998-2999): host path, batched device verdict
    and the oracle's restatement agree run by run.  The scene's only field is far away, the arm folds its elbow until
    the hand reaches the shoulder."""
    import re
    from or_cdchomp_amd import robots
    mod = bindings.bind(_mk())
    model, base, dofvals, adofs = common.wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    mod.add_kinbody_boxes("far", [([0, 0, 0, 0, 0, 0, 1], [0.05, 0.05, 0.05])], transform=[5, 5, 5, 0, 0, 0, 1])
    mod.SendCommand("computedistancefield kinbody far")
    data, lengths, gpose = mod.get_sdf("far")
    grid = oracle.OraGrid(data, lengths)
    wpose = np.array(gpose, dtype=float); wpose[:3] += 5.0
    rob = oracle.OraRobot(model)
    excl = oracle.self_pairs_excluded(rob)
    links = model.arrays()["sphere_link"]
    # the rule: the same link, parent and child, declared adjacent links (wam2 - wam4) and links touching at zero are
    # skipped; the hand against the base sphere or the upper arm is not
    assert excl[links[1], links[2]] == 1 and excl[links[4], links[6]] == 1 and excl[links[0], links[10]] == 0 and excl[links[1], links[9]] == 0
    vmax = np.ones(model.n_dof)
    mod.set_velocity_limits(model.name, vmax)
    n_runs = 24
    rng = np.random.default_rng(41)
    goals = np.tile(np.asarray(robots.WAM_START), (n_runs, 1))
    goals[:, 3] = np.linspace(2.3, 3.05, n_runs)
    goals[:, 2] += rng.uniform(-0.3, 0.3, n_runs); goals[:, 5] += rng.uniform(-0.5, 0.5, n_runs); goals[:, 4] += rng.uniform(-1, 1, n_runs)
    kw = dict(n_points=40, lambda_=100.0, obs_factor=0.0, obs_factor_self=0.0)
    bid = mod.batch_create(model.name, goals, **kw)
    got = mod.batch_collision_verdict(bid)
    trajs = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    n_self = 0
    for k in range(n_runs):
        orun = oracle.OraRun(rob, base, dofvals, adofs, goals[k], [grid], [wpose], oracle.default_params(**kw))
        orun.set_traj(trajs[k])
        want = orun.collision_recheck(vmax[:7])
        orun.destroy()
        assert want["collides"] == got["collides"][k], (k, want, {q: got[q][k] for q in got})
        if want["collides"]:
            assert want["field"] <= -2                                     # nothing but the robot itself is near
            assert want["sphere"] == got["sphere"][k] and want["field"] == got["field"][k], (k, want, got["sphere"][k], got["field"][k])
            assert np.isclose(want["time"], got["time"][k], rtol=1e-12, atol=1e-15)
            assert np.isclose(want["depth"], got["depth"][k], rtol=1e-9, atol=1e-12)
            n_self += 1
        # the single-run command path (host)
        run = mod.create(robot=model.name, adofgoal=list(goals[k]), **kw)
        mod.gettraj(run=run, no_collision_exception=True)
        details = mod.last_collision_details()
        if want["collides"]:
            m = re.match(r"Collision at t=(\S+): spheres (\d+) and (\d+) of \S+ overlap by (\S+) m", details)
            assert m, (k, details)
            assert int(m.group(2)) == want["sphere"] and int(m.group(3)) == -2 - want["field"]
            assert np.isclose(float(m.group(1)), want["time"], rtol=1e-5, atol=1e-9)
            assert np.isclose(float(m.group(4)), want["depth"], rtol=1e-4, atol=1e-9)
            with pytest.raises(RuntimeError, match="Resulting trajectory is in collision!"):
                mod.gettraj(run=run)
            # the stand-in can be left out: for one call (an additive flag of gettraj) ...
            mod.SendCommand("gettraj run %s no_self_collision_check" % run)
            assert mod.last_collision_details() == ""
        else:
            assert details == ""
        mod.destroy(run=run)
    assert 3 <= n_self < n_runs, n_self
    # ... or for the robot (orc_robot_set_self_check): the field leg of the re-check alone is left, and nothing is near the field
    mod.set_self_check(model.name, False)
    bid = mod.batch_create(model.name, goals, **kw)
    assert mod.batch_collision_verdict(bid)["collides"].sum() == 0
    run = mod.create(robot=model.name, adofgoal=list(goals[-1]), **kw)
    mod.gettraj(run=run)
    mod.destroy(run=run)
    mod.set_self_check(model.name, True)
    assert mod.batch_collision_verdict(bid)["collides"].sum() == n_self
    ver = np.full(n_runs, -7, dtype=np.int32); out = np.zeros((n_runs, 40, 7))
    mod.SendCommand("gettrajbatch run %d out 0x%x verdict 0x%x no_self_collision_check" % (bid, out.ctypes.data, ver.ctypes.data))
    assert ver.sum() == 0
    mod.batch_destroy(bid)
