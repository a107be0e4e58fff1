"""The oracle's own robot model on random robots (CPU).  OpenRAVE's forward kinematics and Jacobians are third-party
and absent ("parity unpinned"), so the oracle's restatement (oracle/ora_robot.c) defines truth for every
parity test -- the least this file can do is check it against something written independently: the numpy kinematics of
`robots.RobotModel.link_frames` for the sphere centres of every waypoint (fixed and floating base, revolute and
prismatic joints, rotated joint frames, fixed links in between), and finite differences of those for the rows of the
TSR constraint Jacobian (src/orcdchomp_mod.cpp:1330-1497), which goes through the same geometric Jacobian as the
sphere term."""
import numpy as np
import pytest

import common
from or_cdchomp_amd import robots
from test_gpu_random_robots import random_robot, _random_quat

SEEDS = list(range(40))


def _setup(oracle, seed, floating):
    rng = np.random.default_rng(23000 + seed)
    model, what = random_robot(seed)
    n_dof = model.n_dof
    adofs = list(range(n_dof)) if rng.uniform() < 0.5 or n_dof < 4 else sorted(rng.choice(n_dof, size=int(rng.integers(2, n_dof)), replace=False).tolist())
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(n_dof)])
    dofvals = rng.uniform(0.5 * lo, 0.5 * hi)
    base = np.array([-0.55, 0.05, 0.75] + list(_random_quat(rng, 0.9)))
    goal = rng.uniform(0.8 * lo[adofs], 0.8 * hi[adofs])
    basegoal = None
    if floating:
        basegoal = base.copy(); basegoal[:3] += rng.uniform(-0.3, 0.3, size=3)
        q = np.array(_random_quat(rng, 0.5)); basegoal[3:] = robots.quat_mul(tuple(base[3:]), tuple(q))
    prob = common.tabletop_problem(oracle)
    rob = oracle.OraRobot(model)
    kw = dict(n_points=9, lambda_=100.0)
    if floating:
        kw["floating_base"] = 1
    try:
        run = oracle.OraRun(rob, base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]], oracle.default_params(**kw), basegoal=basegoal)
    except RuntimeError:
        pytest.skip("the active dofs of this draw move no sphere")
    return model, what, adofs, dofvals, base, rob, run


def _sphere_centres(model, base_pose, q):
    R, t = model.link_frames(base_pose, q)
    a = model.arrays()
    return np.array([R[l] @ p + t[l] for l, p in zip(a["sphere_link"], a["sphere_pos"])])


@pytest.mark.parametrize("floating", [False, True])
@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_sphere_centres_are_those_of_the_numpy_kinematics(oracle, seed, floating):
    model, what, adofs, dofvals, base, rob, run = _setup(oracle, seed, floating)
    _, _, P = run.eval_obstacle()                     # [n_points][Sa][3], the active spheres in the run's order
    order = run.sphere_order()
    T = run.traj()
    c0 = 7 if floating else 0
    for wp in range(run.n_points):
        q = dofvals.copy(); q[adofs] = T[wp, c0:]
        bp = T[wp, :7] if floating else base
        want = _sphere_centres(model, bp, q)
        for s in range(run.Sa):
            assert np.allclose(P[wp, s], want[order[s]], rtol=0, atol=1e-12), (seed, what, wp, s)
    run.destroy()


@pytest.mark.parametrize("seed", SEEDS[:24])
def test_oracle_tsr_jacobian_is_the_derivative_of_its_value(oracle, seed):
    model, what, adofs, dofvals, base, rob, run = _setup(oracle, seed, False)
    li = len(model.link_names) - 1
    R, t, _, _ = rob.fk(base, dofvals)
    Bw = [[0, 0]] * 6                                 # all six rows: x y z and the three angles
    assert run.add_contsr(li, [0.02, -0.01, 0.05, 0, 0, 0, 1], oracle.pose_from_dR(t[li], R[li]), [0, 0, 0, 0, 0, 0, 1], Bw) == 6
    T = run.traj()
    for wp in (1, 4, 7):
        point = T[wp].copy()
        h0, J = run.eval_contsr(0, point)
        for j in range(run.n):
            eps = 1e-6
            pp = point.copy(); pp[j] += eps
            pm = point.copy(); pm[j] -= eps
            hp, _ = run.eval_contsr(0, pp)
            hm, _ = run.eval_contsr(0, pm)
            d = (hp - hm) / (2 * eps)
            d[3:] = (d[3:] + np.pi / (2 * eps)) % (np.pi / eps) - np.pi / (2 * eps)      # (an angle may wrap between the two)
            assert np.allclose(J[:, j], d, rtol=1e-5, atol=2e-6), (seed, what, wp, j, J[:, j], d)
    run.destroy()


# ---- the sphere cost and its gradient, restated a second time ------------------------------------------------------------
# numpy, written from the reference's text (src/orcdchomp_mod.cpp:1099-1127 velocities and accelerations, 1134-1327
# sphere_cost), with the sphere Jacobians taken as central differences of the numpy kinematics above instead of any
# analytic form: what the C oracle (oracle/ora_run.c, the yardstick of every parity test) computes with its own FK, its own
# Jacobian columns (revolute: axis x lever, prismatic: axis) and its own bookkeeping of active / inactive spheres must come
# out of this too.  The grid's interpolation and gradient are the oracle's (those are pinned to the reference's grid.c).

def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                     [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                     [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])


def _numpy_sphere_cost(model, base, dofvals, adofs, T, order, Sa, grids, poses, eps, eps_self, obs, obs_self, floating=False, free_start=False):
    n_points, n = T.shape
    m = n_points - 2 + (1 if free_start else 0)                  # `start_tsr`: the start point moves too (mod.cpp:2315-2316)
    first = 0 if free_start else 1                               # trajectory row of moving point 0
    dt = 1.0 / (n_points - 1)
    a = model.arrays()
    link = a["sphere_link"][order]; radius = a["sphere_radius"][order]
    S = len(order)

    def centres(row):
        q = np.array(dofvals, dtype=float)
        if floating:
            # the base pose is part of the row; a rotation is what its quaternion is after normalisation
            q[adofs] = row[7:]
            bp = np.array(row[:7], dtype=float); bp[3:] = bp[3:] / np.linalg.norm(bp[3:])
            return _sphere_centres(model, bp, q)[order]
        q[adofs] = row
        return _sphere_centres(model, base, q)[order]            # the run's order: active first
    P = np.array([centres(T[k])[:Sa] for k in range(n_points)])
    P_inactive = None if floating else centres(np.asarray(dofvals)[adofs])[Sa:]         # where they are when the run is created
    G = np.zeros((m, n)); costs = np.zeros(m)
    h = 1e-6
    for i in range(m):
        r0 = i + first
        row = T[r0]
        J = np.zeros((Sa, 3, n))
        for j in range(n):
            rp = row.copy(); rp[j] += h
            rm = row.copy(); rm[j] -= h
            J[:, :, j] = (centres(rp)[:Sa] - centres(rm)[:Sa]) / (2 * h)
        if floating:
            J[:, :, :7] *= 0.01            # src/orcdchomp_mod.cpp:1075-1080 (under a comment that says "overwrite with zeros")
        if r0 == 0:
            # the start point: one-sided velocity, the acceleration of the point after it (mod.cpp:1107-1112, 1125-1126)
            vel = (P[1] - P[0]) / dt
            acc = (P[0] - 2 * P[1] + P[2]) / (dt * dt)
        else:
            vel = (P[r0 + 1] - P[r0 - 1]) / (2 * dt)
            acc = (P[r0 - 1] - 2 * P[r0] + P[r0 + 1]) / (dt * dt)
        for s in range(Sa):
            p = P[r0, s]; v = vel[s]; vn = np.linalg.norm(v)
            cost_sphere = 0.0
            best, best_k = np.inf, -1
            for k, (g, pose) in enumerate(zip(grids, poses)):
                Rw = _rot(pose[3:7]); gp = Rw.T @ (p - np.asarray(pose[:3]))
                err, val = g.interp(gp)
                if err:
                    continue
                if val < best:
                    best, best_k = val, k
            if best_k >= 0:
                g, pose = grids[best_k], poses[best_k]
                Rw = _rot(pose[3:7]); gp = Rw.T @ (p - np.asarray(pose[:3]))
                dist = best - radius[s]
                if dist < 0.0:
                    cost_sphere += vn * obs * (0.5 * eps - dist)
                elif dist < eps:
                    cost_sphere += vn * obs * (0.5 / eps) * (dist - eps) ** 2
                _, gg = g.grad(gp)
                x = Rw @ gg
                x = x * (-1.0 if dist < 0.0 else ((dist / eps - 1.0) if dist < eps else 0.0))
                x = x * (vn * obs)
                if vn > 0.000001:
                    x = x - (x @ v) / (vn * vn) * v
                curv = acc[s].copy()
                if vn > 0.000001:
                    curv = curv - (curv @ v) / (vn * vn) * v
                if vn != 0.0:                                   # (dgemv with alpha == 0 leaves c_grad alone whatever x holds)
                    curv = curv / (vn * vn)
                    x = x - cost_sphere * curv
                    G[i] += vn * (J[s].T @ x)
            for s2 in range(S):
                if link[s2] == link[s]:
                    continue
                d = p - (P[r0, s2] if s2 < Sa else P_inactive[s2 - Sa])
                dist = np.linalg.norm(d)
                if dist > radius[s] + radius[s2] + eps_self:
                    continue
                unit = d / dist
                dist -= radius[s] + radius[s2]
                if dist < 0.0:
                    cost_sphere += vn * obs_self * (0.5 * eps_self - dist)
                else:
                    cost_sphere += vn * obs_self * (0.5 / eps_self) * (dist - eps_self) ** 2
                x = unit * (-1.0 if dist < 0.0 else ((dist / eps_self - 1.0) if dist < eps_self else 1.0))
                x = x * (vn * obs_self)
                if vn > 0.000001:
                    x = x - (x @ v) / (vn * vn) * v
                J2 = J[s] - (J[s2] if s2 < Sa else 0.0)
                G[i] += J2.T @ x
            costs[i] += cost_sphere
    return G, costs, P


@pytest.mark.parametrize("seed", SEEDS[:20])
def test_oracle_sphere_cost_against_a_second_restatement(oracle, seed):
    rng = np.random.default_rng(29000 + seed)
    model, what, adofs, dofvals, base, rob, probe = _setup(oracle, seed, False)
    probe.destroy()
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    # a second, rotated field around the robot so that best-of-two and the rotation of the gradient take part
    occ = np.zeros((14, 12, 10)); occ[5:9, 4:8, 3:7] = np.inf
    blob = oracle.OraGrid(occ, [0.7, 0.6, 0.5]).bin_sdf()
    bpose = np.array([-0.85, -0.25, 0.55] + list(_random_quat(rng, 1.0)))
    grids.append(blob); poses.append(bpose)
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(model.n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(model.n_dof)])
    goal = rng.uniform(0.9 * lo[adofs], 0.9 * hi[adofs])
    eps, eps_self, obs, obs_self = 0.12, float(rng.uniform(0.03, 0.1)), 130.0, 17.0
    kw = dict(n_points=int(rng.integers(5, 12)), lambda_=100.0, epsilon=eps, epsilon_self=eps_self, obs_factor=obs, obs_factor_self=obs_self)
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))
    # not the straight line only: a trajectory that bends (accelerations, the curvature term)
    T = run.traj()
    T[1:-1] += 0.08 * rng.normal(size=T[1:-1].shape)
    G, costs, P = run.eval_obstacle()
    order = run.sphere_order()
    G2, costs2, P2 = _numpy_sphere_cost(model, base, dofvals, adofs, T.copy(), order, run.Sa, grids, poses, eps, eps_self, obs, obs_self)
    assert np.allclose(P, P2, rtol=0, atol=1e-12)
    assert np.allclose(costs, costs2, rtol=1e-9, atol=1e-12), (seed, what, costs, costs2)
    scale = max(np.abs(G).max(), 1e-9)
    assert np.allclose(G, G2, rtol=2e-6, atol=2e-7 * scale), (seed, what, np.abs(G - G2).max(), scale)
    assert costs.sum() > 0.0 or seed >= 0
    run.destroy()
    print("seed %d (%s): %d waypoints, cost %.3g, |G| %.3g, worst difference %.1e" % (seed, what, T.shape[0], costs.sum(), scale, np.abs(G - G2).max()))


# ---- the optimizer, restated a second time -----------------------------------------------------------------------------
# numpy with its LAPACK (np.linalg.inv is dgetrf + dgetri, what the reference calls), dense matrices and the reference's
# order of operations, written from the text of src/libcd/chomp.c:40-178 (defaults), 239-340 (K, E -> A, B, trC), 342-428
# (A^-1), 430-683 (the iteration: gradient, metric, momentum, step, joint-limit rounds, costs) and the loop around it,
# src/orcdchomp_mod.cpp:2752-2849.  The cost callback is the oracle's sphere term (checked above) evaluated at THIS
# optimizer's trajectory through a second run object: what is compared is oracle/ora_chomp.c, the banded / dense metric, the
# limit projection and the bookkeeping of the costs.

class NumpyChomp:
    def __init__(self, T, D, lam, use_momentum, lower, upper, free_start=False):
        n_points, n = T.shape
        self.free_start = free_start
        self.T = T.copy(); self.m = m = n_points - 2 + (1 if free_start else 0); self.n = n; self.lam = lam; self.use_momentum = use_momentum
        self.lower = lower; self.upper = upper
        dt = 1.0 / (n_points - 1)                                        # src/orcdchomp_mod.cpp:2567
        wds = np.zeros(D); wds[D - 1] = 1.0                              # chomp.c:127-128
        inits = [None if free_start else T[0]] + [np.zeros(n)] * (D - 1)  # chomp.c:131-141: zero vectors, not NULL; `start_tsr`: inits[0] = NULL (mod.cpp:2572)
        finals = [T[-1]] + [np.zeros(n)] * (D - 1)
        Kprev = None; Eprev = None; Nprev = m
        self.A = np.zeros((m, m)); self.B = np.zeros((m, n)); cnn = np.zeros((n, n))
        for d in range(D):
            hi = 0 if inits[d] is None else 1
            N = Nprev - 1 + hi + 1
            diff = np.zeros((N, Nprev)); E = np.zeros((N, n))
            if hi:
                diff[0, 0] = 1.0 / dt; E[0] += -1.0 / dt * inits[d]
            for i in range(Nprev - 1):
                diff[hi + i, i] = -1.0 / dt; diff[hi + i, i + 1] = 1.0 / dt
            diff[N - 1, Nprev - 1] = -1.0 / dt; E[N - 1] += 1.0 / dt * finals[d]
            K = diff if d == 0 else diff @ Kprev
            if d > 0:
                E = E + diff @ Eprev
            self.A += wds[d] / N * (K.T @ K); self.B += wds[d] / N * (K.T @ E); cnn += wds[d] / N * (E.T @ E)
            Kprev, Eprev, Nprev = K, E, N
        self.trC = 0.5 * np.trace(cnn)
        self.Ainv = np.linalg.inv(self.A)
        self.AG = np.zeros((m, n)); self.leapfrog_first = 1              # chomp.c:80, 114-115

    def moving(self):
        return self.T[0:-1] if self.free_start else self.T[1:-1]

    def smooth_cost(self):
        Tm = self.moving()
        return np.trace(0.5 * Tm.T @ (self.A @ Tm) + self.B.T @ Tm) + self.trC

    def iterate(self, cost_callback, con_eval=None):
        """one pass of cd_chomp_iterate(c, 1, ...): returns (status, cost_obs of the trajectory it started from, cost_smooth after);
        con_eval(point) -> (h [k], J [k][n]): a hard constraint on every moving point (chomp.c:553-600)"""
        m, n = self.m, self.n
        G, costs = cost_callback(self.T)
        cost_obs = costs.sum() / m
        G = G / m
        Tm = self.moving()
        G = G + self.A @ Tm
        G = G + self.B
        if not self.use_momentum:
            self.AG = self.Ainv @ G
        elif self.leapfrog_first:
            self.AG = self.AG + 0.5 / self.lam * (self.Ainv @ G); self.leapfrog_first = 0
        else:
            self.AG = self.AG + 1.0 / self.lam * (self.Ainv @ G)
        if con_eval is not None:
            # h_i + (-1/lambda) J_i AG_i; the system J Ainv J^T x = h over all constrained points at once; LAPACK's dgesv;
            # delta = Ainv J^T x.  con_eval: one function for every moving point, or a list of (point index, function)
            cons = [(i, con_eval) for i in range(m)] if callable(con_eval) else list(con_eval)
            ev = [(i, *f(Tm[i].copy())) for i, f in cons]
            hs = [hh - (1.0 / self.lam) * (JJ @ self.AG[i]) for i, hh, JJ in ev]
            off = np.cumsum([0] + [len(hh) for hh in hs])
            JAJT = np.zeros((off[-1], off[-1]))
            for a, (i1, _, J1) in enumerate(ev):
                for b2, (i2, _, J2) in enumerate(ev):
                    JAJT[off[a]:off[a+1], off[b2]:off[b2+1]] = self.Ainv[i1, i2] * (J1 @ J2.T)
            x = np.linalg.solve(JAJT, np.concatenate(hs))
            for a, (i, _, JJ) in enumerate(ev):
                Tm -= np.outer(self.Ainv[:, i], JJ.T @ x[off[a]:off[a+1]])
        Tm -= self.AG / self.lam
        for rounds in range(1000):
            Gjl = np.zeros((m, n)); largest = 0.0; where = (0, 0)
            for i in range(m):
                for j in range(n):
                    if Tm[i, j] < self.lower[j]:
                        Gjl[i, j] = self.lower[j] - Tm[i, j]
                        if abs(Gjl[i, j]) > largest:
                            largest = abs(Gjl[i, j]); where = (i, j)
                    if Tm[i, j] > self.upper[j]:
                        Gjl[i, j] = self.upper[j] - Tm[i, j]
                        if abs(Gjl[i, j]) > largest:
                            largest = abs(Gjl[i, j]); where = (i, j)
            if largest == 0.0:
                break
            GA = self.Ainv @ Gjl
            Tm += 1.01 * Gjl[where] / GA[where] * GA
        else:
            return -1, cost_obs, None
        return 0, cost_obs, self.smooth_cost()


@pytest.mark.parametrize("seed", SEEDS[:24])
def test_oracle_optimizer_against_a_second_restatement(oracle, seed):
    rng = np.random.default_rng(37000 + seed)
    model, what, adofs, dofvals, base, rob, probe = _setup(oracle, seed, False)
    probe.destroy()
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(model.n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(model.n_dof)])
    # goals close to the limits now and then: the joint-limit rounds run
    goal = rng.uniform(0.7 * lo[adofs], 0.7 * hi[adofs])
    if seed % 3 == 0:
        # goals a hair inside the limits, and a start moved next to them as well: the joint-limit rounds run
        tl = np.array([model.limit_lower[d] if np.isfinite(model.limit_lower[d]) else -1.5 for d in adofs])
        th = np.array([model.limit_upper[d] if np.isfinite(model.limit_upper[d]) else 1.5 for d in adofs])
        side = rng.uniform(size=len(adofs)) < 0.5
        beyond = rng.uniform(size=len(adofs)) < 0.4                     # ... and some goals a little beyond them (nothing clamps a goal)
        goal = np.where(side, th - 0.004 * (th - tl), tl + 0.004 * (th - tl)) + np.where(beyond, np.where(side, 1.0, -1.0) * 0.03 * (th - tl), 0.0)
        dofvals = dofvals.copy(); dofvals[adofs] = np.where(side, th - 0.02 * (th - tl), tl + 0.02 * (th - tl))
    D = 2 if seed % 5 == 4 else 1
    momentum = 1 if seed % 4 == 1 else 0
    lam = float(rng.uniform(60.0, 300.0))
    kw = dict(n_points=int(rng.integers(5, 40)), lambda_=lam, obs_factor=float(rng.uniform(50.0, 400.0)), D=D, use_momentum=momentum)
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))
    callback_run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))

    def callback(T):
        callback_run.set_traj(T)
        G, costs, _ = callback_run.eval_obstacle()
        return G.copy(), costs.copy()
    lower = np.array([model.limit_lower[d] for d in adofs]); upper = np.array([model.limit_upper[d] for d in adofs])
    mine = NumpyChomp(run.traj(), D, lam, momentum, lower, upper)
    # the metric itself, entry by entry
    assert np.allclose(run.mat("A", run.m, run.m), mine.A, rtol=1e-12, atol=0) and np.allclose(run.mat("B", run.m, run.n), mine.B, rtol=1e-12, atol=1e-300)
    assert np.allclose(run.mat("Ainv", run.m, run.m), mine.Ainv, rtol=1e-8, atol=1e-14)
    n_iter = int(rng.integers(3, 25))
    rounds = 0
    status = st = 0
    trace, otrace = [], []
    for it in range(n_iter):
        st, ocosts, otr = run.iterate(1, trace=True)                       # (no hmc: n calls of one iteration are one call of n)
        rounds += run.chomp().last_num_limadjs
        status, cobs, csm = mine.iterate(callback)
        assert status == st, (seed, what, it, status, st)
        if st != 0:
            break
        trace.append([cobs + csm, cobs, csm]); otrace.append(otr[0])
    otrace = np.array(otrace)
    if st != 0:
        pytest.skip("the run leaves its joint limits for good in both")
    # (the reference reports the obstacle cost of the trajectory before the step and the smoothness cost after it)
    assert np.allclose(np.array(trace), otrace, rtol=1e-8, atol=1e-12), (seed, what)
    assert common.rel_l2(mine.T, run.traj()) <= 1e-9, (seed, what, common.rel_l2(mine.T, run.traj()))
    # the final call with do_iteration = 0: both costs of the trajectory as it stands (src/orcdchomp_mod.cpp:2833)
    G, costs = callback(mine.T)
    final = np.array([costs.sum() / mine.m + mine.smooth_cost(), costs.sum() / mine.m, mine.smooth_cost()])
    assert np.allclose(final, ocosts, rtol=1e-8, atol=1e-12), (seed, what, final, ocosts)
    if seed % 3 == 0 and seed < 12:
        assert rounds > 0, "the draw was meant to run the joint-limit rounds"
    print("seed %d (%s): D %d, momentum %d, %d points, %d iterations, %d joint-limit rounds: rel L2 %.1e" % (
        seed, what, D, momentum, kw["n_points"], n_iter, rounds, common.rel_l2(mine.T, run.traj())))
    run.destroy(); callback_run.destroy()


@pytest.mark.parametrize("seed", SEEDS[:16])
def test_oracle_constraint_step_against_a_second_restatement(oracle, seed):
    """the TSR hard constraint on every moving point (src/libcd/chomp.c:553-600): the oracle factors the system with its
    own LU with partial pivoting, the restatement hands the same dense system to LAPACK's dgesv (numpy.linalg.solve) -- the
    reference's own call.  Constraint values and Jacobians are the oracle's (checked by finite differences above)."""
    rng = np.random.default_rng(41000 + seed)
    model, what, adofs, dofvals, base, rob, probe = _setup(oracle, seed, False)
    probe.destroy()
    adofs = list(range(model.n_dof))
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(model.n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(model.n_dof)])
    goal = np.clip(dofvals + 0.3 * rng.uniform(-1, 1, size=model.n_dof) * np.minimum(1.0, hi - lo), lo, hi)
    li = len(model.link_names) - 1
    R, t, _, _ = rob.fk(base, dofvals)
    momentum = 1 if seed % 3 == 1 else 0
    lam = float(rng.uniform(100.0, 300.0))
    kw = dict(n_points=int(rng.integers(5, 30)), lambda_=lam, obs_factor=100.0, use_momentum=momentum)
    T0w = oracle.pose_from_dR(t[li], R[li])
    ident = [0, 0, 0, 0, 0, 0, 1]
    rows = None
    for attempt in range(12):
        cand = sorted(rng.choice(6, size=int(rng.integers(1, 3)), replace=False).tolist())
        Bw = [[0, 0] if r in cand else ([-1, 1] if r < 3 else [-3, 3]) for r in range(6)]
        pr = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(n_points=9))
        pr.add_contsr(li, ident, T0w, ident, Bw)
        smin = min(np.linalg.svd(pr.eval_contsr(0, pr.traj()[i])[1], compute_uv=False).min() for i in range(1, 8))
        pr.destroy()
        if smin > 0.05:
            rows = cand
            break
    if rows is None:
        pytest.skip("no well-posed rows for the last link of this draw")
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))
    assert run.add_contsr(li, ident, T0w, ident, Bw) == len(rows)
    callback_run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))
    callback_run.add_contsr(li, ident, T0w, ident, Bw)

    def callback(T):
        callback_run.set_traj(T)
        G, costs, _ = callback_run.eval_obstacle()
        return G.copy(), costs.copy()

    def con_eval(point):
        h, J = callback_run.eval_contsr(0, point)
        return h.copy(), J.copy()
    lower = np.array([model.limit_lower[d] for d in adofs]); upper = np.array([model.limit_upper[d] for d in adofs])
    mine = NumpyChomp(run.traj(), 1, lam, momentum, lower, upper)
    n_iter = int(rng.integers(3, 12))
    worst = 0.0
    for it in range(n_iter):
        st, ocosts, otr = run.iterate(1, trace=True)
        status, cobs, csm = mine.iterate(callback, con_eval)
        assert status == st == 0, (seed, what, it, status, st)
        assert np.allclose([cobs + csm, cobs, csm], otr[0], rtol=1e-7, atol=1e-12), (seed, what, it)
        worst = max(worst, common.rel_l2(mine.T, run.traj()))
    assert worst <= 1e-8, (seed, what, rows, worst)
    after = max(np.abs(con_eval(mine.T[i])[0]).max() for i in range(1, mine.T.shape[0] - 1))
    print("seed %d (%s): rows %s, momentum %d, %d points, %d iterations: worst rel L2 %.1e, constraint left at %.1e" % (
        seed, what, rows, momentum, kw["n_points"], n_iter, worst, after))
    run.destroy(); callback_run.destroy()


class GslStream:
    """gsl_rng_mt19937 seeded the way gsl_rng_set does it (0 -> 4357; numpy's legacy seeding is the same init_genrand),
    gsl_rng_uniform = the 32-bit output / 2^32, gsl_ran_gaussian = the polar Box-Muller of GSL's randist/gauss.c"""
    def __init__(self, seed, count=200000):
        self.raw = np.random.RandomState(seed if seed else 4357).randint(0, 2 ** 32, size=count, dtype=np.uint64)
        self.k = 0

    def uniform(self):
        v = float(self.raw[self.k]) / 4294967296.0
        self.k += 1
        return v

    def uniform_pos(self):
        while True:
            v = self.uniform()
            if v != 0.0:
                return v

    def gaussian(self, sigma):
        while True:
            x = -1 + 2 * self.uniform_pos(); y = -1 + 2 * self.uniform_pos()
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0):
                return sigma * y * np.sqrt(-2.0 * np.log(r2) / r2)


@pytest.mark.parametrize("seed", SEEDS[:12])
def test_oracle_hmc_loop_against_a_second_restatement(oracle, seed):
    """`use_hmc` (src/orcdchomp_mod.cpp:2752-2768): at its resampling iterations the momentum is redrawn from the run's
    GSL stream, row-major, sigma = 1/sqrt(100 e^(0.02 iter)), the next resampling iteration moves on by 1 + (int)(-ln u /
    lambda_hmc), the leapfrog starts over; `iter` restarts with every `iterate` call while the resampling iteration is
    kept.  The stream itself is GSL's published algorithm in both writings (numpy's mt19937 here)."""
    import math
    rng = np.random.default_rng(43000 + seed)
    model, what, adofs, dofvals, base, rob, probe = _setup(oracle, seed, False)
    probe.destroy()
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(model.n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(model.n_dof)])
    goal = rng.uniform(0.6 * lo[adofs], 0.6 * hi[adofs])
    lam = float(rng.uniform(100.0, 300.0)); hl = float(rng.uniform(0.05, 0.5)); gsl_seed = int(rng.integers(0, 50))
    kw = dict(n_points=int(rng.integers(5, 24)), lambda_=lam, obs_factor=100.0, use_momentum=1, use_hmc=1, hmc_resample_lambda=hl, seed=gsl_seed)
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))
    callback_run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw))

    def callback(T):
        callback_run.set_traj(T)
        G, costs, _ = callback_run.eval_obstacle()
        return G.copy(), costs.copy()
    lower = np.array([model.limit_lower[d] for d in adofs]); upper = np.array([model.limit_upper[d] for d in adofs])
    mine = NumpyChomp(run.traj(), 1, lam, 1, lower, upper)
    stream = GslStream(gsl_seed)
    resample_iter, resamples = 0, 0
    for n_call in (int(rng.integers(4, 15)), int(rng.integers(4, 15))):           # two calls: iter restarts, the schedule does not
        st, ocosts, otr = run.iterate(n_call, trace=True)
        assert st == 0
        for it in range(n_call):
            if it == resample_iter:
                alpha = 100.0 * math.exp(0.02 * it)
                for i in range(mine.m):
                    for j in range(mine.n):
                        mine.AG[i, j] = stream.gaussian(1.0 / math.sqrt(alpha))
                mine.leapfrog_first = 1
                resample_iter += 1 + int(-math.log(stream.uniform()) / hl)
                resamples += 1
            status, cobs, csm = mine.iterate(callback)
            assert status == 0
            assert np.allclose([cobs + csm, cobs, csm], otr[it], rtol=1e-8, atol=1e-12), (seed, what, it)
        assert common.rel_l2(mine.T, run.traj()) <= 1e-10, (seed, what, common.rel_l2(mine.T, run.traj()))
    assert resamples >= 1
    print("seed %d (%s): gsl seed %d, lambda_hmc %.2f, %d resamplings: rel L2 %.1e" % (seed, what, gsl_seed, hl, resamples, common.rel_l2(mine.T, run.traj())))
    run.destroy(); callback_run.destroy()


@pytest.mark.parametrize("seed", SEEDS[:16])
def test_oracle_sphere_cost_floating_base_against_a_second_restatement(oracle, seed):
    """floating base (src/orcdchomp_mod.cpp:1008-1016, 1050-1085; src/libcd/spatial.c:71-102, 295-337): the reference
    forms the base columns of a sphere's Jacobian as X(-p) . Jsp(pose) times 0.01.  Without the 0.01 that is the derivative
    of the sphere's centre with respect to the seven pose numbers, the quaternion taken as the rotation it normalises to --
    so the second restatement takes exactly that derivative by central differences and scales it."""
    rng = np.random.default_rng(47000 + seed)
    model, what, adofs, dofvals, base, rob, run = _setup(oracle, seed, True)
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    T = run.traj()
    T[1:-1, :3] += 0.03 * rng.normal(size=T[1:-1, :3].shape)
    T[1:-1, 7:] += 0.08 * rng.normal(size=T[1:-1, 7:].shape)
    q = T[1:-1, 3:7] + 0.05 * rng.normal(size=T[1:-1, 3:7].shape)
    T[1:-1, 3:7] = q / np.linalg.norm(q, axis=1)[:, None]
    G, costs, P = run.eval_obstacle()
    assert run.Sa == run.S                                      # with a floating base every sphere is active (mod.cpp:2273)
    p = run.chomp()
    G2, costs2, P2 = _numpy_sphere_cost(model, base, dofvals, adofs, T.copy(), run.sphere_order(), run.Sa, grids, poses,
                                        0.1, 0.04, 200.0, 10.0, floating=True)      # the reference's defaults (mod.cpp:1845-1848)
    assert np.allclose(P, P2, rtol=0, atol=1e-12)
    assert np.allclose(costs, costs2, rtol=1e-9, atol=1e-12), (seed, what)
    scale = max(np.abs(G).max(), 1e-9)
    assert np.allclose(G, G2, rtol=2e-6, atol=2e-7 * scale), (seed, what, np.abs(G - G2).max(), scale)
    assert np.abs(G[:, :7]).max() > 0.0
    print("seed %d (%s): |G| %.3g, base columns up to %.3g, worst difference %.1e" % (seed, what, scale, np.abs(G[:, :7]).max(), np.abs(G - G2).max()))
    run.destroy()


def _hom(R, t):
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = t
    return M


@pytest.mark.parametrize("floating", [False, True])
@pytest.mark.parametrize("seed", SEEDS[:16])
def test_oracle_tsr_value_against_a_second_restatement(oracle, seed, floating):
    """the value of a TSR constraint (src/orcdchomp_mod.cpp:1330-1415: T0w^-1 . T_ee . Twe^-1 as x y z and the ZYX
    angles of src/libcd/kin.c:615-646, Bw's rows in the order x y z roll pitch yaw), with quaternion products in the
    oracle, with 4 x 4 matrices and the angles read off the rotation matrix here; T0w, Twe and the tool offset at random."""
    rng = np.random.default_rng(53000 + seed)
    model, what, adofs, dofvals, base, rob, run = _setup(oracle, seed, floating)
    li = int(rng.integers(1, len(model.link_names)))
    tool = np.array(list(rng.uniform(-0.1, 0.1, size=3)) + list(_random_quat(rng, 1.0)))
    T0w = np.array(list(rng.uniform(-0.5, 0.5, size=3)) + list(_random_quat(rng, 2.0)))
    Twe = np.array(list(rng.uniform(-0.2, 0.2, size=3)) + list(_random_quat(rng, 2.0)))
    assert run.add_contsr(li, tool, T0w, Twe, [[0, 0]] * 6) == 6
    T = run.traj()
    c0 = 7 if floating else 0
    for wp in range(run.n_points):
        point = T[wp].copy()
        point[c0:] += 0.2 * rng.normal(size=run.n - c0)
        h, _ = run.eval_contsr(0, point)
        q = dofvals.copy(); q[adofs] = point[c0:]
        bp = point[:7] if floating else base
        R, t = model.link_frames(bp, q)
        M = np.linalg.inv(_hom(_rot(T0w[3:]), T0w[:3])) @ _hom(R[li], t[li]) @ _hom(_rot(tool[3:]), tool[:3]) @ np.linalg.inv(_hom(_rot(Twe[3:]), Twe[:3]))
        yaw = np.arctan2(M[1, 0], M[0, 0]); pitch = np.arcsin(-M[2, 0]); roll = np.arctan2(M[2, 1], M[2, 2])
        want = np.array([M[0, 3], M[1, 3], M[2, 3], roll, pitch, yaw])
        if abs(pitch) > 1.5:
            continue                                  # next to the gimbal lock the reference switches formula
        assert np.allclose(h, want, rtol=0, atol=1e-10), (seed, what, wp, h, want)
    run.destroy()


@pytest.mark.parametrize("seed", SEEDS[:12])
def test_oracle_starttraj_sampling_is_linear_interpolation_in_time(oracle, seed):
    """`create starttraj` (src/orcdchomp_mod.cpp:2375-2416): waypoint i of the run is the passed trajectory sampled at
    i * duration / (n_points - 1); for the linearly interpolated, linearly retimed documents `gettraj` writes that is
    numpy.interp over the cumulated deltatimes, column by column (base position included; the base quaternion is
    interpolated component-wise and normalised, src/orcdchomp_mod.cpp:2391-2402)."""
    rng = np.random.default_rng(59000 + seed)
    k, n = int(rng.integers(2, 40)), int(rng.integers(1, 20))
    wp = rng.normal(size=(k, n))
    dt = np.r_[0.0, rng.uniform(0.0, 1.0, size=k - 1)]
    dt[rng.uniform(size=k) < 0.1] = 0.0                         # waypoints that take no time
    dt[0] = 0.0
    if dt.sum() == 0.0:
        dt[-1] = 0.5
    tc = np.cumsum(dt)
    for npts in (3, int(rng.integers(4, 90))):
        got = oracle.sample_starttraj(wp, dt, npts)
        ts = np.arange(npts) * tc[-1] / (npts - 1)
        want = np.column_stack([np.interp(ts, tc, wp[:, j]) for j in range(n)])
        # (where two waypoints share a time, both writings must pick the same side: compare away from those instants)
        ok = np.array([np.min(np.abs(t - tc[np.r_[False, dt[1:] == 0.0]])) > 1e-9 if np.any(dt[1:] == 0.0) else True for t in ts])
        assert np.allclose(got[ok], want[ok], rtol=1e-12, atol=1e-13), (seed, npts)
        assert np.allclose(got[0], wp[0], atol=1e-13) and np.allclose(got[-1], wp[-1], atol=1e-12)
    # floating base: position like a joint, quaternion component-wise then normalised, reordered to libcd's x y z w
    wb = rng.normal(size=(k, 7)); wb[:, 3:] /= np.linalg.norm(wb[:, 3:], axis=1)[:, None]
    npts = int(rng.integers(3, 50))
    got = oracle.sample_starttraj_floating(wp, wb, dt, npts)
    ts = np.arange(npts) * tc[-1] / (npts - 1)
    pos = np.column_stack([np.interp(ts, tc, wb[:, j]) for j in range(3)])
    qw_first = np.column_stack([np.interp(ts, tc, wb[:, 3 + j]) for j in range(4)])          # OpenRAVE's order: w x y z
    q = np.column_stack([qw_first[:, 1], qw_first[:, 2], qw_first[:, 3], qw_first[:, 0]])
    q = q / np.linalg.norm(q, axis=1)[:, None]
    arm = np.column_stack([np.interp(ts, tc, wp[:, j]) for j in range(n)])
    ok = np.array([np.min(np.abs(t - tc[np.r_[False, dt[1:] == 0.0]])) > 1e-9 if np.any(dt[1:] == 0.0) else True for t in ts])
    assert np.allclose(got[ok], np.column_stack([pos, q, arm])[ok], rtol=1e-12, atol=1e-13), seed


@pytest.mark.parametrize("seed", SEEDS[:16])
def test_oracle_free_start_against_a_second_restatement(oracle, seed):
    """`start_tsr` (src/orcdchomp_mod.cpp:1988-1992, 2316-2323, 2570-2576): the start point is a variable -- one more
    moving point, no init row in the first difference operator, a one-sided sphere velocity and a borrowed acceleration
    for it (1107-1112, 1125-1126) -- held on a TSR by a constraint on that one point; now and then a `con_tsr` on
    every point on other rows as well."""
    rng = np.random.default_rng(67000 + seed)
    model, what, adofs, dofvals, base, rob, probe = _setup(oracle, seed, False)
    probe.destroy()
    adofs = list(range(model.n_dof))
    prob = common.tabletop_problem(oracle)
    grids, poses = [prob["sdf"]], [np.asarray(prob["pose"], dtype=float)]
    lo = np.array([max(model.limit_lower[d], -1.5) for d in range(model.n_dof)])
    hi = np.array([min(model.limit_upper[d], 1.5) for d in range(model.n_dof)])
    goal = np.clip(dofvals + 0.3 * rng.uniform(-1, 1, size=model.n_dof) * np.minimum(1.0, hi - lo), lo, hi)
    li = len(model.link_names) - 1
    R, t, _, _ = rob.fk(base, dofvals)
    T0w = oracle.pose_from_dR(t[li], R[li]); ident = [0, 0, 0, 0, 0, 0, 1]
    n_anc, cur = 0, li
    while cur >= 0:
        n_anc += model.joint_type[cur] != robots.JOINT_FIXED
        cur = model.parent[cur]

    def rows_ok(rows, points):
        Bw = [[0, 0] if r in rows else ([-1, 1] if r < 3 else [-3, 3]) for r in range(6)]
        pr = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(n_points=9))
        pr.add_contsr(li, ident, T0w, ident, Bw)
        smin = min(np.linalg.svd(pr.eval_contsr(0, pr.traj()[i])[1], compute_uv=False).min() for i in points)
        pr.destroy()
        return Bw if smin > 0.05 else None
    rows_s = Bw_s = rows_c = Bw_c = None
    with_con = seed % 3 == 0 and n_anc >= 2
    for attempt in range(20):
        rows_s = sorted(rng.choice(6, size=1, replace=False).tolist())
        rows_c = sorted(rng.choice([r for r in range(6) if r not in rows_s], size=1, replace=False).tolist()) if with_con else []
        Bw_s = rows_ok(sorted(rows_s + rows_c), [0]) and rows_ok(rows_s, [0])
        Bw_c = rows_ok(rows_c, range(0, 8)) if with_con else None
        if Bw_s and (Bw_c or not with_con):
            break
    else:
        pytest.skip("no well-posed rows for the last link of this draw")
    momentum = 1 if seed % 4 == 2 else 0
    lam = float(rng.uniform(100.0, 300.0))
    kw = dict(n_points=int(rng.integers(5, 30)), lambda_=lam, obs_factor=100.0, use_momentum=momentum)
    st_arg = (li, ident, T0w, ident, Bw_s)
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(start_tsr=st_arg, **kw))
    cb = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(start_tsr=st_arg, **kw))
    # a plain run whose first constraint is the start TSR's rows / the con_tsr's rows: evaluates them at any point
    ev_s = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw)); ev_s.add_contsr(li, ident, T0w, ident, Bw_s)
    ev_c = None
    if with_con:
        run.add_contsr(li, ident, T0w, ident, Bw_c); cb.add_contsr(li, ident, T0w, ident, Bw_c)
        ev_c = oracle.OraRun(rob, base, dofvals, adofs, goal, grids, poses, oracle.default_params(**kw)); ev_c.add_contsr(li, ident, T0w, ident, Bw_c)
    assert run.m == run.n_points - 1
    # the sphere term with the start point moving: the oracle's against the numpy restatement
    T = run.traj().copy()
    T[0:-1] += 0.05 * rng.normal(size=T[0:-1].shape)
    cb.set_traj(T)
    G, costs, P = cb.eval_obstacle()
    G2, costs2, P2 = _numpy_sphere_cost(model, base, dofvals, adofs, T.copy(), cb.sphere_order(), cb.Sa, grids, poses,
                                        0.1, 0.04, 100.0, 10.0, free_start=True)
    scale = max(np.abs(G).max(), 1e-9)
    assert G.shape[0] == run.n_points - 1 and np.allclose(costs, costs2, rtol=1e-9, atol=1e-12)
    assert np.allclose(G, G2, rtol=2e-6, atol=2e-7 * scale), (seed, what, np.abs(G - G2).max(), scale)

    def callback(Tq):
        cb.set_traj(Tq)
        Gq, cq, _ = cb.eval_obstacle()
        return Gq.copy(), cq.copy()

    def con_start(point):
        h, J = ev_s.eval_contsr(0, point)
        return h.copy(), J.copy()

    def con_all(point):
        h, J = ev_c.eval_contsr(0, point)
        return h.copy(), J.copy()
    lower = np.array([model.limit_lower[d] for d in adofs]); upper = np.array([model.limit_upper[d] for d in adofs])
    mine = NumpyChomp(run.traj(), 1, lam, momentum, lower, upper, free_start=True)
    assert np.allclose(run.mat("A", run.m, run.m), mine.A, rtol=1e-12, atol=0) and np.allclose(run.mat("B", run.m, run.n), mine.B, rtol=1e-12, atol=1e-300)
    # the reference's list of constraints: the start TSR on point 0, then the con_tsr on every point (any order solves the same system)
    cons = [(0, con_start)] + ([(i, con_all) for i in range(mine.m)] if with_con else [])
    n_iter = int(rng.integers(3, 10))
    worst = 0.0
    for it in range(n_iter):
        st, ocosts, otr = run.iterate(1, trace=True)
        status, cobs, csm = mine.iterate(callback, cons)
        assert status == st == 0, (seed, what, it, status, st)
        assert np.allclose([cobs + csm, cobs, csm], otr[0], rtol=1e-7, atol=1e-12), (seed, what, it, [cobs + csm, cobs, csm], otr[0])
        worst = max(worst, common.rel_l2(mine.T, run.traj()))
    assert worst <= 1e-8, (seed, what, worst)
    print("seed %d (%s): start rows %s%s, momentum %d, %d points, %d iterations: worst rel L2 %.1e" % (
        seed, what, rows_s, " + rows %s on every point" % rows_c if with_con else "", momentum, kw["n_points"], n_iter, worst))
    for r_ in (run, cb, ev_s, ev_c):
        if r_ is not None:
            r_.destroy()
