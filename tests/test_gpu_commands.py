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


@pytest.mark.parametrize("n_points", [3, 4, 7, 65, 130])
def test_trajectory_lengths(oracle, n_points):
    """m = 1 (a single moving waypoint) up to more waypoints than one tile holds"""
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
