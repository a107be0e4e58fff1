"""-m gpu: behaviour at the edges of the iterate boundary: a run that leaves its joint limits (per-run
status, costs, trace, a later call), create's dat_filename log, iterate's max_time and
trajs_fileformstr with and without hmc, lane placement independent of the batch, one batch cut over
several devices inside one process."""
import os
import re

import numpy as np
import pytest

import common
import or_cdchomp_amd
from or_cdchomp_amd import bindings

pytestmark = pytest.mark.gpu

KW = dict(n_points=100, lambda_=100.0, obs_factor=500.0)


def _mk(devices=0):
    import or_cdchomp_amd
    return or_cdchomp_amd.Module(devices)


def test_run_outside_joint_limits_keeps_going_in_a_later_call(oracle):
    """reference: the iterate call throws (src/orcdchomp_mod.cpp:2799-2803), the run stays usable.  Here:
    status -1 for that call, iterations made, costs of the last complete iteration, NaN trace rows for
    the iterations not made, and the next call iterates the run again"""
    mod = _mk()
    model = common.setup_product_wam(mod)
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    goals = common.wam_goals(256, seed=20250101)
    bid = mod.batch_create(model.name, goals, **KW)
    costs, status = mod.batch_iterate(bid, 100)
    iters = mod.batch_iterations_done(bid)
    trace = mod.batch_trace(bid, 100)
    t1 = mod.batch_gettraj(bid)
    bad = np.where(status == -1)[0]
    assert len(bad) > 0, "the workload is expected to contain runs that leave their limits"
    assert (iters[status == 0] == 100).all() and (iters[bad] < 100).all()
    rob = oracle.OraRobot(model)
    same_iter = 0
    for k in bad:
        assert np.isfinite(trace[k, :iters[k]]).all() and np.isnan(trace[k, iters[k]:]).all()
        if iters[k] > 0:
            assert np.array_equal(costs[k], trace[k, iters[k]-1])      # the last complete iteration's costs
        run = oracle.OraRun(rob, base, dofvals, adofs, goals[k], [prob["sdf"]], [prob["pose"]], oracle.default_params(**KW))
        st, _ = run.iterate(100)
        same_iter += int(st == -1 and run.iter() == iters[k])
        run.destroy()
    # such runs are chaotic shortly before they fail: most, not all, fail in the oracle's iteration
    assert same_iter >= 0.6 * len(bad), (same_iter, len(bad))
    costs2, status2 = mod.batch_iterate(bid, 10)
    iters2 = mod.batch_iterations_done(bid)
    t2 = mod.batch_gettraj(bid)
    ok = status == 0
    assert (iters2[ok & (status2 == 0)] == 10).all()
    for k in bad:
        assert not np.array_equal(t1[k], t2[k])                          # not frozen: iterated (or projected) again
    assert np.isfinite(costs2[status2 == 0]).all()
    # the runs that were fine are what a fresh 110-iteration batch makes of them
    ref = mod.batch_create(model.name, goals, **KW)
    _, s110 = mod.batch_iterate(ref, 110)
    t110 = mod.batch_gettraj(ref)
    good = ok & (status2 == 0) & (s110 == 0)
    assert np.array_equal(t2[good], t110[good])


def test_dat_filename_log(oracle, tmp_path):
    """create dat_filename: "%d %f %f %f %f\\n" per iteration = iteration, seconds, total, obs, smooth
    (src/orcdchomp_mod.cpp:2306-2310, 2811-2818); the iteration counter restarts with every call"""
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    goal = common.wam_goals(1, seed=31)[0]
    path = str(tmp_path / "chomp.dat")
    kw = dict(n_points=60, lambda_=100.0, obs_factor=500.0)
    run = mod.create(robot=model.name, adofgoal=list(goal), dat_filename=path, **kw)
    mod.iterate(run=run, n_iter=5)
    mod.iterate(run=run, n_iter=3, max_time=1e9)          # the one-iteration-per-launch path writes the same lines
    mod.destroy(run=run)
    rows = [ln.split() for ln in open(path).read().splitlines()]
    assert len(rows) == 8 and all(len(r) == 5 for r in rows)
    assert all(re.fullmatch(r"-?\d+\.\d{6}", v) for r in rows for v in r[1:])
    assert [int(r[0]) for r in rows] == [0, 1, 2, 3, 4, 0, 1, 2]
    secs = np.array([float(r[1]) for r in rows])
    assert (np.diff(secs[:5]) > 0).all() and (np.diff(secs[5:]) > 0).all() and secs[0] > 0
    orun = oracle.OraRun(oracle.OraRobot(model), base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]],
                         oracle.default_params(**kw))
    _, _, tr1 = orun.iterate(5, trace=True)
    _, _, tr2 = orun.iterate(3, trace=True)
    want = np.concatenate([tr1, tr2])
    got = np.array([[float(v) for v in r[2:]] for r in rows])
    assert np.allclose(got, want, rtol=1e-6, atol=6e-7)                 # %f keeps six decimals
    # a batch takes a pattern with the run index
    goals = common.wam_goals(3, seed=32)
    g = np.ascontiguousarray(goals)
    bid = int(mod.SendCommand("createbatch robot %s n_runs 3 adofgoals 0x%x n_points 60 lambda 100 dat_filename '%s'"
                              % (model.name, g.ctypes.data, str(tmp_path / "run%d.dat"))))
    mod.SendCommand("iteratebatch run %d n_iter 4" % bid)
    mod.batch_destroy(bid)
    for k in range(3):
        assert len(open(str(tmp_path / ("run%d.dat" % k))).read().splitlines()) == 4


@pytest.mark.parametrize("hmc", [0, 1])
def test_iterate_max_time_and_trajs_fileformstr(oracle, tmp_path, hmc):
    """iterate ... trajs_fileformstr FMT max_time T (src/orcdchomp_mod.cpp:2752-2828): the trajectory is
    dumped BEFORE every iteration, the loop stops after the iteration that crosses max_time; with
    use_hmc the resample schedule follows the call's own iteration counter across the launches"""
    mod = bindings.bind(_mk())
    model = common.setup_product_wam(mod)
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    goal = common.wam_goals(1, seed=41)[0]
    kw = dict(n_points=50, lambda_=100.0, obs_factor=500.0)
    extra = dict(use_momentum=1, use_hmc=1, hmc_resample_lambda=0.3, seed=7) if hmc else {}
    run = mod.create(robot=model.name, adofgoal=list(goal), **dict(kw, **extra))
    fmt = str(tmp_path / "traj_%03d.xml")
    cost = [None]
    mod.iterate(run=run, n_iter=12, trajs_fileformstr=fmt, max_time=1e9, cost=cost)
    final = mod.batch_gettraj(int(run))[0]
    files = sorted(os.listdir(str(tmp_path)))
    assert files == ["traj_%03d.xml" % k for k in range(12)]
    rob = oracle.OraRobot(model)
    okw = dict(kw, **extra)

    def oracle_after(k):
        r = oracle.OraRun(rob, base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]], oracle.default_params(**okw))
        st, c = r.iterate(k)
        t = r.traj().copy()
        r.destroy()
        return t, c

    for k in (0, 1, 5, 11):
        dumped = bindings.parse_traj(open(fmt % k).read())
        assert dumped.shape == (50, 7)
        assert common.rel_l2(dumped, oracle_after(k)[0]) <= 1e-6, k
    t12, c12 = oracle_after(12)
    assert common.rel_l2(final, t12) <= 1e-6
    assert np.isclose(cost[0], c12[0], rtol=1e-5)          # `sout << cost_total`: six significant digits (mod.cpp:2849)
    mod.destroy(run=run)
    # the launches of that path against ONE fused launch of the same call
    a = mod.create(robot=model.name, adofgoal=list(goal), **dict(kw, **extra))
    b = mod.create(robot=model.name, adofgoal=list(goal), **dict(kw, **extra))
    mod.iterate(run=a, n_iter=9)
    mod.iterate(run=b, n_iter=9, max_time=1e9)
    assert np.array_equal(mod.batch_gettraj(int(a)), mod.batch_gettraj(int(b)))
    # max_time 0: the check comes after the iteration, so exactly one is made
    c = mod.create(robot=model.name, adofgoal=list(goal), **dict(kw, **extra))
    mod.iterate(run=c, n_iter=9, max_time=0.0)
    assert common.rel_l2(mod.batch_gettraj(int(c))[0], oracle_after(1)[0]) <= 1e-6


def test_hmc_device_streams_across_launches(monkeypatch):
    """the device-resident mt19937 streams (batches of 256 runs and more) keep their place across the
    launches of one iterate call: iteratebatch with max_time equals the fused call"""
    monkeypatch.setenv("ORC_HMC_DEVICE", "1")
    mod = _mk()
    model = common.setup_product_wam(mod)
    goals, basegoals, seeds, kw = common.config4_problem(8)
    kw = dict(kw, n_points=40, hmc_resample_lambda=0.25)
    a = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
    b = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
    mod.batch_iterate(a, 14)
    mod.SendCommand("iteratebatch run %d n_iter 14 max_time 1e9" % b)
    assert np.array_equal(mod.batch_gettraj(a), mod.batch_gettraj(b))
    # and a second call continues both alike (hmc_resample_iter persists, the counter restarts)
    mod.batch_iterate(a, 6)
    mod.SendCommand("iteratebatch run %d n_iter 6 max_time 1e9" % b)
    assert np.array_equal(mod.batch_gettraj(a), mod.batch_gettraj(b))


def test_results_do_not_depend_on_batch_composition(oracle, monkeypatch):
    """the lane placement of the spheres fixes the order in which pair forces are summed; it is a
    function of the robot alone, so a run's bits cannot depend on which batch a module saw first"""
    goals = common.wam_goals(64, seed=51)
    others_a = common.wam_goals(200, seed=52)
    others_b = np.tile(common.wam_goals(1, seed=53), (40, 1))            # a very different first batch
    trajs = []
    for first in (others_a, others_b):
        mod = _mk()                                                      # fresh module: fresh placement cache
        model = common.setup_product_wam(mod)
        warm = mod.batch_create(model.name, first, **KW)
        mod.batch_iterate(warm, 2)
        mixed = np.concatenate([first[:7], goals])
        bid = mod.batch_create(model.name, mixed, **KW)
        mod.batch_iterate(bid, 100)
        trajs.append(mod.batch_gettraj(bid)[7:])
    assert np.array_equal(trajs[0], trajs[1])
    # two different placements (the annealed one and the sorted order) differ only by the rounding of
    # a different summation order
    monkeypatch.setenv("ORC_NO_PLACEMENT", "1")
    mod = _mk()
    model = common.setup_product_wam(mod)
    bid = mod.batch_create(model.name, goals, **KW)
    _, st = mod.batch_iterate(bid, 100)
    plain = mod.batch_gettraj(bid)
    prob = common.tabletop_problem(oracle)
    _, base, dofvals, adofs = common.wam_state()
    ora = lambda g: oracle.batch_run(oracle.OraRobot(model), base, dofvals, adofs, g, [prob["sdf"]], [prob["pose"]],
                                     oracle.default_params(**KW), 100)
    ores = ora(goals)
    otraj, ost = ores[0], ores[2]
    amps, _ = common.amplification(ora, goals, ores)
    errs = []
    for k in range(64):
        if st[k] != 0 or ost[k] != 0:
            continue
        amp = amps[k]
        e = common.rel_l2(plain[k], trajs[0][k])
        errs.append(e)
        assert e <= max(1e-9, common.CHAOS_FACTOR * amp), (k, e, amp)
        assert common.rel_l2(plain[k], otraj[k]) <= max(1e-6, common.CHAOS_FACTOR * amp)
    assert np.median(errs) <= 1e-12


def test_one_batch_over_two_shards_in_one_process(monkeypatch):
    """orc_module_new_multi / `createbatch ... devices 'i j'`: contiguous blocks per device, results
    gathered into the caller's arrays; on this box both shards sit on GPU 0 (own streams).  hmc seeds,
    warm starts, traces and the collision verdict follow the cut"""
    monkeypatch.setenv("ORC_HMC_DEVICE", "1")
    one = _mk(0)
    two = _mk([0, 0, 0])
    model = common.setup_product_wam(one)
    common.setup_product_wam(two)
    goals, basegoals, seeds, kw = common.config4_problem(37)             # uneven cut: 13 + 12 + 12
    kw = dict(kw, n_points=50, hmc_resample_lambda=0.2)
    res = []
    for mod in (one, two):
        bid = mod.batch_create(model.name, goals, basegoals=basegoals, seeds=seeds, **kw)
        c, s = mod.batch_iterate(bid, 20)
        res.append(dict(c=c, s=s, t=mod.batch_gettraj(bid), tr=mod.batch_trace(bid, 20), it=mod.batch_iterations_done(bid),
                        ag=mod.batch_state(bid, "AG"), v=mod.batch_collision_verdict(bid)))
        warm = mod.batch_gettraj(bid)
        mod.batch_set_traj(bid, warm[::-1].copy())                       # a warm start that crosses the cut
        assert np.array_equal(mod.batch_gettraj(bid), warm[::-1])
    a, b = res
    for key in ("c", "s", "t", "tr", "it", "ag"):
        assert np.array_equal(a[key], b[key]), key
    for key in a["v"]:
        assert np.array_equal(a["v"][key], b["v"][key]), key
    # the command form on a one-device module
    g = np.ascontiguousarray(goals[:, :]); bg = np.ascontiguousarray(basegoals); sd = np.ascontiguousarray(seeds)
    out = np.zeros_like(a["t"])
    bid = int(one.SendCommand("createbatch robot %s n_runs 37 adofgoals 0x%x basegoals 0x%x seeds 0x%x floating_base use_momentum "
                              "use_hmc hmc_resample_lambda 0.2 n_points 50 lambda 100 obs_factor 500 devices '0 0'"
                              % (model.name, g.ctypes.data, bg.ctypes.data, sd.ctypes.data)))
    one.SendCommand("iteratebatch run %d n_iter 20" % bid)
    one.SendCommand("gettrajbatch run %d out 0x%x" % (bid, out.ctypes.data))
    assert np.array_equal(out, a["t"])
    with pytest.raises(RuntimeError, match="bad device ordinal"):
        one.SendCommand("createbatch robot %s n_runs 37 adofgoals 0x%x basegoals 0x%x floating_base devices '0 99'"
                        % (model.name, g.ctypes.data, bg.ctypes.data))


def test_workgroup_shape_is_a_module_setting(oracle):
    """orc_set_workgroup_threads: 192-thread workgroups (four per CU) give the trajectories of the default
    shape (the same arithmetic per waypoint and per column; only the cost sums are grouped differently),
    and the shape of a module's batches does not depend on their size"""
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    goals = common.wam_goals(96, seed=77)
    kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
    a = mod.batch_create(model.name, goals, **kw)
    ca, sa = mod.batch_iterate(a, 60)
    ta = mod.batch_gettraj(a)
    mod.batch_destroy(a)
    mod.set_workgroup_threads(192)
    b = mod.batch_create(model.name, goals, **kw)
    cb, sb = mod.batch_iterate(b, 60)
    tb = mod.batch_gettraj(b)
    # a subset in a batch of its own: bit for bit the same under the module's shape
    c = mod.batch_create(model.name, goals[10:30], **kw)
    cc, sc = mod.batch_iterate(c, 60)
    tc = mod.batch_gettraj(c)
    mod.batch_destroy(b); mod.batch_destroy(c)
    mod.set_workgroup_threads(0)
    assert np.array_equal(sa, sb)
    assert np.array_equal(ta, tb)
    assert np.allclose(ca, cb, rtol=1e-13, atol=0)
    assert np.array_equal(tc, tb[10:30]) and np.array_equal(cc, cb[10:30]) and np.array_equal(sc, sb[10:30])
    with pytest.raises(RuntimeError, match="workgroup threads must be 0"):
        mod.set_workgroup_threads(160)
    # 128 threads: two wavefronts on a run, eight runs per CU (the shape the planner gives TSR-constrained runs of a module whose
    # launches overlap): the same trajectories
    mod.set_workgroup_threads(128)
    h = mod.batch_create(model.name, goals[:40], **kw)
    ch, sh = mod.batch_iterate(h, 60)
    th = mod.batch_gettraj(h)
    mod.batch_destroy(h)
    mod.set_workgroup_threads(0)
    assert np.array_equal(sh, sa[:40]) and np.array_equal(th, ta[:40])
    assert np.allclose(ch, ca[:40], rtol=1e-13, atol=0)
    # the latency shape (eight wavefronts on one run, the whole trajectory in one tile): the same trajectories
    mod.set_workgroup_threads(512)
    d = mod.batch_create(model.name, goals[:24], **kw)
    cd, sd = mod.batch_iterate(d, 60)
    td = mod.batch_gettraj(d)
    mod.batch_destroy(d)
    mod.set_workgroup_threads(0)
    assert np.array_equal(sd, sa[:24]) and np.array_equal(td, ta[:24])
    assert np.allclose(cd, ca[:24], rtol=1e-13, atol=0)
    # orc_set_workgroups_per_cu(4): the kernels built for four 256-thread workgroups per CU (128 registers, three tiles
    # for the WAM instead of two): the same trajectories; a robot the budget is not built for keeps its default
    mod.set_workgroups_per_cu(4)
    e = mod.batch_create(model.name, goals, **kw)
    ce, se = mod.batch_iterate(e, 60)
    te = mod.batch_gettraj(e)
    mod.batch_destroy(e)
    f = mod.batch_create(model.name, goals[:8], floating_base=1, basegoals=np.tile(common.wam_state()[1], (8, 1)), **kw)
    mod.batch_iterate(f, 3)
    mod.batch_destroy(f)
    # ... and so does a run the budget has no room for (800 waypoints: the trajectory alone is 45 KB, a workgroup's share 40 KB)
    long_kw = dict(kw, n_points=800)
    g4 = mod.batch_create(model.name, goals[:4], **long_kw)
    cg4, sg4 = mod.batch_iterate(g4, 5)
    tg4 = mod.batch_gettraj(g4)
    mod.batch_destroy(g4)
    mod.set_workgroups_per_cu(0)
    g0 = mod.batch_create(model.name, goals[:4], **long_kw)
    cg0, sg0 = mod.batch_iterate(g0, 5)
    assert np.array_equal(tg4, mod.batch_gettraj(g0)) and np.array_equal(cg4, cg0) and np.array_equal(sg4, sg0)
    mod.batch_destroy(g0)
    assert np.array_equal(se, sa) and np.array_equal(te, ta)
    assert np.allclose(ce, ca, rtol=1e-13, atol=0)
    with pytest.raises(RuntimeError, match="workgroups per CU must be 0"):
        mod.set_workgroups_per_cu(5)
    # what the caller does not say the planner chooses from the MODULE's settings: overlapping launches (orc_set_num_streams >= 2)
    # take the four-per-CU kernels; 3 says "the kernels' own budget" explicitly.  The trajectories do not notice.
    mod2 = or_cdchomp_amd.Module(0)
    common.setup_product_wam(mod2)
    mod2.set_num_streams(2)
    for budget in (0, 3, 4):
        mod2.set_workgroups_per_cu(budget)
        i = mod2.batch_create(model.name, goals[:32], **kw)
        ci, si = mod2.batch_iterate(i, 60)
        assert np.array_equal(si, sa[:32]) and np.array_equal(mod2.batch_gettraj(i), ta[:32]), budget
        assert np.allclose(ci, ca[:32], rtol=1e-13, atol=0)
        mod2.batch_destroy(i)
    mod2.close()


def test_chunked_iterate_of_a_batch_keeps_aborted_runs_out(oracle, tmp_path):
    """iterate with max_time / trajs_fileformstr runs one launch per iteration; a run of a BATCH that leaves its joint
    limits in one of them must stay out for the rest of the call, as it does in the fused path (and as the reference,
    which has thrown by then, src/orcdchomp_mod.cpp:2799-2803): same final trajectories, status and iterations made"""
    mod = _mk()
    model = common.setup_product_wam(mod)
    goals = common.wam_goals(256, seed=20250101)
    a = mod.batch_create(model.name, goals, **KW)
    b = mod.batch_create(model.name, goals, **KW)
    _, st_a = mod.batch_iterate(a, 60)
    costs = np.zeros((256, 3)); st_b = np.full(256, 9, dtype=np.int32)
    mod.SendCommand("iteratebatch run %d n_iter 60 max_time 1e9 costs 0x%x status 0x%x" % (b, costs.ctypes.data, st_b.ctypes.data))
    assert (st_a != 0).sum() >= 3, "the workload should hold a few runs that leave their joint limits"
    assert np.array_equal(st_a, st_b)
    ia, ib = mod.batch_iterations_done(a), mod.batch_iterations_done(b)
    assert np.array_equal(ia, ib) and ia[st_a != 0].max() < 60 and (ia[st_a == 0] == 60).all()
    assert np.array_equal(mod.batch_gettraj(a), mod.batch_gettraj(b))
    mod.batch_destroy(a); mod.batch_destroy(b)


def test_file_patterns_are_checked_before_they_reach_printf(tmp_path):
    """dat_filename of a batch: exactly one integer conversion (the run); trajs_fileformstr: the iteration, and the run
    for a batch; anything else in the pattern (a %s, a %n, a missing conversion) is "Bad arguments!" """
    mod = _mk()
    model = common.setup_product_wam(mod)
    g = np.ascontiguousarray(common.wam_goals(2, seed=3))
    create = "createbatch robot %s n_runs 2 adofgoals 0x%x n_points 20 " % (model.name, g.ctypes.data)
    for bad in ("x.dat", "x_%s.dat", "x_%d_%d.dat", "x_%n.dat", "x_%f.dat"):
        with pytest.raises(RuntimeError, match="Bad arguments!"):
            mod.SendCommand(create + "dat_filename '%s'" % str(tmp_path / bad))
    bid = int(mod.SendCommand(create + "dat_filename '%s'" % str(tmp_path / "ok_%02d_100%%.dat")))
    mod.SendCommand("iteratebatch run %d n_iter 2" % bid)
    assert os.path.exists(str(tmp_path / "ok_00_100%.dat")) and os.path.exists(str(tmp_path / "ok_01_100%.dat"))
    for bad in ("t_%d.xml", "t.xml", "t_%d_%s.xml"):
        with pytest.raises(RuntimeError, match="Bad arguments!"):
            mod.SendCommand("iteratebatch run %d n_iter 1 trajs_fileformstr '%s'" % (bid, str(tmp_path / bad)))
    mod.SendCommand("iteratebatch run %d n_iter 2 trajs_fileformstr '%s'" % (bid, str(tmp_path / "t_%03d_r%d.xml")))
    assert os.path.exists(str(tmp_path / "t_001_r1.xml"))
    mod.batch_destroy(bid)
    run = int(mod.SendCommand("create robot %s adofgoal '%s' n_points 20" % (model.name, " ".join("%r" % v for v in g[0]))))
    for bad in ("s_%d_%d.xml", "s_%s.xml"):
        with pytest.raises(RuntimeError, match="Bad arguments!"):
            mod.SendCommand("iterate run %d n_iter 1 trajs_fileformstr '%s'" % (run, str(tmp_path / bad)))
    mod.SendCommand("iterate run %d n_iter 1 trajs_fileformstr '%s'" % (run, str(tmp_path / "s_%d.xml")))
    assert os.path.exists(str(tmp_path / "s_0.xml"))
    # a constant name for one run, as the reference's sprintf takes it: every iteration overwrites the file
    mod.SendCommand("iterate run %d n_iter 2 trajs_fileformstr '%s'" % (run, str(tmp_path / "s.xml")))
    assert os.path.exists(str(tmp_path / "s.xml"))
