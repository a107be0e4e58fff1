"""-m gpu: the WAM of BASELINE configs[1] HOLDING a four-sphere box, at the configuration's full size (1024 runs, 100 waypoints,
100 iterations): the 32-lane kernel family with the dense self-collision pair list (csrc/cost_pairs.h; reference
src/orcdchomp_mod.cpp:2168-2300 for the held body's spheres, 1134-1327 for the cost).

At this size the oracle runs a sample (16 runs: 1e-6 relative L2, north_star) and the rest are properties the domain offers: a run
does not depend on what shares its batch, two calls are one, the costs are the costs of the returned trajectory, the register
budgets and workgroup shapes the family is built for give the same trajectories bit for bit."""
import numpy as np
import pytest

import common
import or_cdchomp_amd

pytestmark = pytest.mark.gpu

KW = dict(common.CONFIG2_KW)
N_RUNS = 1024


@pytest.fixture(scope="module")
def held():
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    goals = common.wam_goals(N_RUNS, seed=20250101)
    bid = mod.batch_create(model.name, goals, **KW)
    costs, status = mod.batch_iterate(bid, 100)
    traj = mod.batch_gettraj(bid)
    dims = mod.batch_dims(bid)
    mod.batch_destroy(bid)
    yield dict(mod=mod, model=model, hand=hand, pose=pose, goals=goals, costs=costs, status=status, traj=traj, dims=dims)
    mod.close()


def test_sample_matches_the_oracle(held, oracle):
    """16 of the 1024 runs against the oracle (the held body through ora_robot.grabbed), the chaotic ones held to their measured
    amplification (tests/common.py)"""
    h = held
    idx = np.unique(np.linspace(0, N_RUNS - 1, 16).astype(int))
    _, base, dofvals, adofs = common.wam_state()
    prob = common.tabletop_problem(oracle)
    rob = oracle.OraRobot(h["model"], grabbed=[(h["hand"], h["pose"], common.HELD4_POS, common.HELD4_RAD)])
    ora = lambda g: oracle.batch_run(rob, base, dofvals, adofs, g, [prob["sdf"]], [prob["pose"]], oracle.default_params(**KW), 100)
    res = ora(h["goals"][idx])
    amp, stable = common.amplification(ora, h["goals"][idx], res)
    moved = ~stable
    otraj, ocosts, ost = res[0], res[1], res[2]
    st = h["status"][idx]
    err = np.array([common.rel_l2(h["traj"][k], otraj[j]) for j, k in enumerate(idx)])
    well = (ost == 0) & (st == 0) & (amp < 1e-9) & ~moved
    assert well.sum() >= 10, (amp, ost, st)
    assert err[well].max() <= 1e-6, err
    assert np.allclose(h["costs"][idx][well], ocosts[well], rtol=1e-6, atol=0)
    ill = (ost == 0) & (st == 0) & ~well
    assert (err[ill] <= np.maximum(1e-6, common.CHAOS_FACTOR * amp[ill])).all(), (err[ill], amp[ill])
    # a status that differs belongs to a run the oracle itself moves under a one-ulp change of its goal
    assert all(amp[j] >= 1e-9 or moved[j] for j in np.flatnonzero(ost != st)), (ost, st, amp)
    print("held4, 16 of 1024 runs vs the oracle: worst rel L2 %.2e over %d well-conditioned runs; %d of 1024 runs outside their limits"
          % (err[well].max(), well.sum(), (h["status"] != 0).sum()))


def test_runs_are_independent_of_batch_and_position(held):
    """a permuted subset, launched on its own, reproduces the full batch bit for bit (the pair list and its summation order are a
    function of the robot, not of the batch)"""
    h = held; mod = h["mod"]
    pick = np.random.default_rng(5).permutation(N_RUNS)[:160]
    bid = mod.batch_create(h["model"].name, h["goals"][pick], **KW)
    costs, status = mod.batch_iterate(bid, 100)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    assert np.array_equal(status, h["status"][pick])
    assert np.array_equal(traj, h["traj"][pick])
    assert np.array_equal(costs, h["costs"][pick])
    # ... and once more in the same order: the launch is deterministic
    bid = mod.batch_create(h["model"].name, h["goals"], **KW)
    c2, s2 = mod.batch_iterate(bid, 100)
    t2 = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    assert np.array_equal(t2, h["traj"]) and np.array_equal(c2, h["costs"]) and np.array_equal(s2, h["status"])


@pytest.mark.parametrize("momentum", [0, 1])
def test_two_calls_are_one(held, momentum):
    h = held; mod = h["mod"]
    goals = h["goals"][:384]
    kw = dict(KW, use_momentum=momentum)
    a = mod.batch_create(h["model"].name, goals, **kw)
    ca, sa = mod.batch_iterate(a, 100)
    ta = mod.batch_gettraj(a)
    b = mod.batch_create(h["model"].name, goals, **kw)
    _, sb1 = mod.batch_iterate(b, 41)
    cb, sb2 = mod.batch_iterate(b, 59)
    tb = mod.batch_gettraj(b)
    mod.batch_destroy(a); mod.batch_destroy(b)
    sb = np.minimum(sb1, sb2)
    ok = (sa == 0) & (sb == 0)
    assert np.array_equal(sa, sb)
    assert np.array_equal(ta[ok], tb[ok])
    assert np.array_equal(ca[ok], cb[ok])


def test_costs_are_the_costs_of_the_returned_trajectory(held):
    h = held; mod = h["mod"]
    goals = h["goals"][:256]
    bid = mod.batch_create(h["model"].name, goals, **KW)
    c1, s1 = mod.batch_iterate(bid, 100)
    t1 = mod.batch_gettraj(bid)
    c2, s2 = mod.batch_iterate(bid, 0)
    t2 = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    ok = s1 == 0
    assert np.array_equal(t1, t2)
    assert np.array_equal(c1[ok], c2[ok])
    assert np.allclose(c1[ok, 0], c1[ok, 1] + c1[ok, 2], rtol=1e-15, atol=0)
    lo = np.asarray(h["model"].limit_lower[:7]); hi = np.asarray(h["model"].limit_upper[:7])
    assert ((t1[ok] >= lo - 1e-12) & (t1[ok] <= hi + 1e-12)).all()


def test_budgets_and_shapes_give_the_same_trajectories(held):
    """the kernels built for four workgroups per CU (orc_set_workgroups_per_cu(4)) and the latency shape (512 threads, one run per
    CU: what the single-run `create` command uses) walk the same pair list in the same order: the same bits"""
    h = held
    goals = h["goals"][:96]
    mod = or_cdchomp_amd.Module(0)
    model, _, _ = common.setup_product_wam_held4(mod)
    out = {}
    for name, wgs, threads in (("default", 0, 0), ("four per CU", 4, 0), ("latency shape", 0, 512)):
        mod.set_workgroups_per_cu(wgs); mod.set_workgroup_threads(threads)
        bid = mod.batch_create(model.name, goals, **KW)
        c, s = mod.batch_iterate(bid, 100)
        out[name] = (mod.batch_gettraj(bid), c, s)
        mod.batch_destroy(bid)
    mod.close()
    assert np.array_equal(out["default"][0], h["traj"][:96]) and np.array_equal(out["default"][2], h["status"][:96])
    for name in ("four per CU", "latency shape"):
        assert np.array_equal(out[name][2], out["default"][2]), name
        assert np.array_equal(out[name][0], out["default"][0]), name
        assert np.allclose(out[name][1], out["default"][1], rtol=1e-13, atol=0), name      # (the cost sums are grouped by wavefront)


def test_the_held_body_changes_the_answer_and_release_restores_it(held):
    """grab -> release gives back the robot that never held anything, bit for bit; holding changes the trajectories"""
    h = held
    goals = h["goals"][:64]
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    bid = mod.batch_create(model.name, goals, **KW)
    mod.batch_iterate(bid, 100)
    bare = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.add_kinbody_boxes("held4", [([0.045, 0.045, 0.01, 0, 0, 0, 1], [0.08, 0.08, 0.04])], transform=h["pose"])
    mod.set_kinbody_spheres("held4", common.HELD4_POS, common.HELD4_RAD)
    mod.grab(model.name, "held4", h["hand"])
    mod.release(model.name, "held4")
    bid = mod.batch_create(model.name, goals, **KW)
    mod.batch_iterate(bid, 100)
    again = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    assert np.array_equal(bare, again)
    ok = h["status"][:64] == 0
    assert max(common.rel_l2(bare[k], h["traj"][k]) for k in np.flatnonzero(ok)) > 1e-4


def test_two_fields_take_the_general_kind(held, oracle):
    """with a second field in the scene (the mug's own) the family's general kind runs (any number of fields, best-of-N lookup,
    src/orcdchomp_mod.cpp:1171-1196) instead of the one-aligned-field kind: 12 runs against the oracle with both fields"""
    h = held
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    mod.SendCommand("computedistancefield kinbody mug aabb_padding 0.15")
    goals = h["goals"][:12]
    kw = dict(n_points=60, lambda_=100.0, obs_factor=500.0)
    bid = mod.batch_create(model.name, goals, **kw)
    costs, status = mod.batch_iterate(bid, 50)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    fields, poses = [], []
    for name in ("table", "mug"):
        data, lengths, fpose = mod.get_sdf(name)
        fields.append(oracle.OraGrid(data, list(lengths))); poses.append(list(fpose))     # (the product's own fields: built bit-identically to the oracle's, tests/test_gpu_sdf_fuzz.py)
    mod.close()
    _, base, dofvals, adofs = common.wam_state()
    rob = oracle.OraRobot(model, grabbed=[(hand, pose, common.HELD4_POS, common.HELD4_RAD)])
    ora = lambda g: oracle.batch_run(rob, base, dofvals, adofs, g, fields, poses, oracle.default_params(**kw), 50)
    res = ora(goals)
    amp, stable = common.amplification(ora, goals, res)
    otraj, ocosts, ost = res[0], res[1], res[2]
    well = (ost == 0) & (status == 0) & (amp < 1e-9) & stable
    err = np.array([common.rel_l2(traj[k], otraj[k]) for k in range(len(goals))])
    assert well.sum() >= 8, (amp, ost, status)
    assert err[well].max() <= 1e-6, err
    assert np.allclose(costs[well], ocosts[well], rtol=1e-6, atol=0)


# ---- round 6: the family beyond the fp64 chain (review item 3; reference src/orcdchomp_mod.cpp:2168-2300, 1251-1317) ----

def _finger_goals(n_runs, seed):
    """the arm's goals of config 2 plus goals for the three finger dofs (inside their limits)"""
    arm = common.wam_goals(n_runs, seed=seed)
    fingers = np.random.default_rng(seed + 7).uniform(0.2, 2.2, size=(n_runs, 3))
    return np.ascontiguousarray(np.hstack([arm, fingers]))


def test_wam_with_finger_dofs_holding_the_box(oracle):
    """A TREE robot in the pair-list family: the WAM with its three finger dofs active (the joint tree forks at the hand) holding the
    four-sphere box, 1024 runs x 100 waypoints x 100 iterations; a 16-run sample against the oracle, and the property that a subset on
    its own equals the same runs inside the batch bit for bit.  `plan` says which kernels ran: variant bit 512 (pair list) and bit 1 (tree)."""
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    adofs = list(range(10))
    mod.set_active_dofs(model.name, adofs)
    goals = _finger_goals(N_RUNS, 20250101)
    bid = mod.batch_create(model.name, goals, **KW)
    plan = mod.batch_plan(bid)
    assert common.plan_switches_active() or (plan["variant"] & 512 and plan["variant"] & 1 and plan["lanes_per_waypoint"] == 32), plan
    costs, status = mod.batch_iterate(bid, 100)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    idx = np.unique(np.linspace(0, N_RUNS - 1, 16).astype(int))
    _, base, dofvals, _ = common.wam_state()
    prob = common.tabletop_problem(oracle)
    rob = oracle.OraRobot(model, grabbed=[(hand, pose, common.HELD4_POS, common.HELD4_RAD)])
    ora = lambda g: oracle.batch_run(rob, base, dofvals, adofs, g, [prob["sdf"]], [prob["pose"]], oracle.default_params(**KW), 100)
    res = ora(goals[idx])
    amp, stable = common.amplification(ora, goals[idx], res)
    otraj, ocosts, ost = res[0], res[1], res[2]
    st = status[idx]
    err = np.array([common.rel_l2(traj[k], otraj[j]) for j, k in enumerate(idx)])
    well = (ost == 0) & (st == 0) & (amp < 1e-9) & stable
    assert well.sum() >= 10, (amp, ost, st)
    assert err[well].max() <= 1e-6, err
    assert np.allclose(costs[idx][well], ocosts[well], rtol=1e-6, atol=0)
    ill = (ost == 0) & (st == 0) & ~well
    # (a run the oracle itself moves by more than the parity bar under a one-ulp change of its goal -- rounding amplified 1e10-fold:
    # ten finger-and-arm dofs bouncing off their limits -- has no digit left that a comparison could hold; it is only asked to stay
    # finite.  One of the sixteen sample runs is of that kind: amplification 1e-4.)
    lost = ill & (amp >= 1e-6)
    held_to = ill & ~lost
    assert (err[held_to] <= np.maximum(1e-6, common.CHAOS_FACTOR * amp[held_to])).all(), (err[held_to], amp[held_to])
    assert np.isfinite(traj[idx][lost]).all() and lost.sum() <= 2, (err[lost], amp[lost])
    assert all(amp[j] >= 1e-9 or not stable[j] for j in np.flatnonzero(ost != st)), (ost, st, amp)
    # a permuted subset on its own
    pick = np.random.default_rng(9).permutation(N_RUNS)[:96]
    bid = mod.batch_create(model.name, goals[pick], **KW)
    c2, s2 = mod.batch_iterate(bid, 100)
    t2 = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    assert np.array_equal(s2, status[pick]) and np.array_equal(t2, traj[pick]) and np.array_equal(c2, costs[pick])
    print("held4 with finger dofs (tree), 16 of 1024 runs vs the oracle: worst rel L2 %.2e over %d well-conditioned runs" % (err[well].max(), well.sum()))


@pytest.mark.parametrize("fingers", [0, 1])
def test_held_box_fp32(oracle, fingers):
    """precision 32 in the pair-list family (chain and tree): 12 runs against the oracle at the fp32 bar of BASELINE configs[4] (1e-3),
    at least 10 of them well-conditioned"""
    mod = or_cdchomp_amd.Module(0)
    model, hand, pose = common.setup_product_wam_held4(mod)
    adofs = list(range(10)) if fingers else list(range(7))
    mod.set_active_dofs(model.name, adofs)
    goals = (_finger_goals(12, 31) if fingers else common.wam_goals(12, seed=31))
    goals[:, :7] = 0.6 * goals[:, :7] + 0.4 * np.asarray(common.wam_state()[2][:7])      # (towards the start: fewer runs at their limits)
    kw = dict(n_points=100, lambda_=100.0, obs_factor=200.0)
    bid = mod.batch_create(model.name, goals, precision=32, **kw)
    plan = mod.batch_plan(bid)
    assert common.plan_switches_active() or (plan["variant"] & 512 and bool(plan["variant"] & 1) == bool(fingers)), plan
    costs, status = mod.batch_iterate(bid, 50)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    _, base, dofvals, _ = common.wam_state()
    prob = common.tabletop_problem(oracle)
    rob = oracle.OraRobot(model, grabbed=[(hand, pose, common.HELD4_POS, common.HELD4_RAD)])
    ora = lambda g: oracle.batch_run(rob, base, dofvals, adofs, g, [prob["sdf"]], [prob["pose"]], oracle.default_params(**kw), 50)
    res = ora(goals)
    amp, stable = common.amplification(ora, goals, res)
    otraj, ocosts, ost = res[0], res[1], res[2]
    # fp32 meets the reference's discontinuities (the one-sided field interpolation picks its neighbour cell by `p < centre`, the range
    # tests, the limit rounds) at a rounding of 6e-8 instead of 1e-16: 1-2 % of such runs leave the fp64 trajectory by more than 1e-4
    # in EITHER fp32 family (pair list 10 of 512, many-sphere 6 of 512: scripts/diag/fp32_pairs_stats.py, profiles/r06_fp32_pairs_stats.txt),
    # which a one-ulp fp64 experiment does not see.  The bar: at least 10 of the 12 runs within 1e-3 (review item 3), none off by more
    # than a trajectory of the same problem can be.
    ok = (ost == 0) & (status == 0) & (amp < 1e-9) & stable
    err = np.array([common.rel_l2(traj[k], otraj[k]) for k in range(len(goals))])
    well = ok & (err <= 1e-3)
    assert ok.sum() >= 11, (amp, ost, status)
    assert well.sum() >= 10, err
    assert err[ok].max() <= 0.1, err
    assert np.allclose(costs[well], ocosts[well], rtol=1e-1, atol=0)
    print("held4 fp32 (%s): worst rel L2 %.2e over %d well-conditioned runs" % ("tree" if fingers else "chain", err[well].max(), well.sum()))


def test_tree_and_chain_kinds_against_the_many_sphere_family(monkeypatch):
    """ORC_PAIRS_CHAIN64_ONLY=1 sends the tree back to the many-sphere family (cost_generic.h, the round-5 path): same trajectories
    to rounding -- two independent implementations of the same sums"""
    if common.plan_switches_active():
        pytest.skip("an experiment switch is set: the comparison is between the two families the planner would choose without it")
    out = {}
    for only in (0, 1):
        if only:
            monkeypatch.setenv("ORC_PAIRS_CHAIN64_ONLY", "1")
        mod = or_cdchomp_amd.Module(0)
        model, hand, pose = common.setup_product_wam_held4(mod)
        mod.set_active_dofs(model.name, list(range(10)))
        goals = _finger_goals(48, 77)
        bid = mod.batch_create(model.name, goals, n_points=60, lambda_=100.0, obs_factor=500.0)
        plan = mod.batch_plan(bid)
        assert bool(plan["variant"] & 512) == (not only), plan
        c, s = mod.batch_iterate(bid, 40)
        out[only] = (mod.batch_gettraj(bid), c, s)
        mod.batch_destroy(bid)
        mod.close()
    ok = (out[0][2] == 0) & (out[1][2] == 0)
    assert ok.sum() >= 30
    errs = np.array([common.rel_l2(out[0][0][k], out[1][0][k]) for k in np.flatnonzero(ok)])
    assert np.median(errs) <= 1e-10, np.sort(errs)[-5:]
