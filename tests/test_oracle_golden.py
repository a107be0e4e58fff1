"""not gpu: the oracle (and the product's host numerics, which need no GPU) against the golden
vectors generated from the reference's own compiled code (tests/golden/make_golden.py), against
oracle/_ref live when it is present, and against the known answers of SURVEY.md 8(c)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import common

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "grid_golden.npz"))


def _host():
    from or_cdchomp_amd import _capi
    return _capi.lib()


def test_bin_sdf_bit_exact(oracle, gold):
    g = oracle.OraGrid(gold["occ"], gold["lengths"])
    assert np.array_equal(g.bin_sdf().data, gold["sdf"])


def test_product_bin_sdf_bit_exact(gold):
    L = _host()
    occ = np.ascontiguousarray(gold["occ"]); out = np.zeros_like(occ)
    sizes = np.asarray(occ.shape, dtype=np.int32); lengths = np.ascontiguousarray(gold["lengths"])
    rc = L.orc_host_bin_sdf(sizes.ctypes.data_as(C.POINTER(C.c_int)), lengths.ctypes.data_as(C.POINTER(C.c_double)),
                            occ.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    assert np.array_equal(out, gold["sdf"])


def test_interp_grad_bit_exact(oracle, gold):
    for data, lengths, pts, vals, grads, errs in (
            (gold["sdf"], gold["lengths"], gold["pts"], gold["sdf_vals"], gold["sdf_grads"], gold["sdf_errs"]),
            (gold["poison"], gold["plen"], gold["ppts"], gold["p_vals"], gold["p_grads"], gold["p_errs"])):
        g = oracle.OraGrid(data, lengths)
        assert (errs != 0).sum() > 0 or data.shape == (6, 7, 8)
        for p, v, gr, e in zip(pts, vals, grads, errs):
            err, val = g.interp(p)
            assert err == e
            if e == 0:
                assert val == v or (np.isinf(val) and np.isinf(v))
                _, gg = g.grad(p)
                fin = np.isfinite(gr)
                assert np.array_equal(gg[fin], gr[fin])
                assert not np.isfinite(gg[~fin]).any()


def test_flood_fill(oracle, gold):
    g = oracle.OraGrid(gold["ff_in"], [1, 1, 1])
    g.flood_fill(0)
    assert np.array_equal(g.data, gold["ff_out"])
    assert g.data[6, 3, 3] == 1.0          # sealed cell untouched
    L = _host()
    cells = np.ascontiguousarray(gold["ff_in"]); sizes = np.asarray(cells.shape, dtype=np.int32)
    assert L.orc_host_flood_fill(sizes.ctypes.data_as(C.POINTER(C.c_int)), cells.ctypes.data_as(C.POINTER(C.c_double)), 0) == 0
    assert np.array_equal(cells, gold["ff_out"])


def _oracle_shparse(oracle, s):
    buf = C.create_string_buffer(s.encode())
    argc = C.c_int(); argv = C.POINTER(C.c_char_p)()
    oracle.lib().ora_util_shparse(buf, C.byref(argc), C.byref(argv))
    return [argv[i].decode() for i in range(argc.value)]


def _product_shparse(s):
    out = C.create_string_buffer(4096)
    n = _host().orc_host_shparse(s.encode(), out, len(out))
    assert n >= 0
    toks = out.raw.split(b"\0")[:n]
    return [t.decode() for t in toks]


def test_shparse(oracle):
    cases = json.load(open(os.path.join(GOLD, "shparse_golden.json")))
    for c in cases:
        assert _oracle_shparse(oracle, c["in"]) == c["tokens"], c["in"]
        assert _product_shparse(c["in"]) == c["tokens"], c["in"]


def test_survey_known_answers(oracle):
    ka = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    L = oracle.lib()
    # grid probe
    gk = ka["grid"]
    occ = np.zeros(gk["sizes"]); occ[tuple(gk["obstacle"])] = np.inf
    g = oracle.OraGrid(occ, [s * gk["cell"] for s in gk["sizes"]]).bin_sdf()
    assert abs(g.data[0, 0, 0] - gk["sdf[0]"]) < 1e-6 and g.data[tuple(gk["obstacle"])] == gk["sdf[obs]"]
    err, v = g.interp(gk["p"]); _, gr = g.grad(gk["p"])
    assert err == 0 and abs(v - gk["interp"]) < 1e-9 and np.allclose(gr, gk["grad"], atol=1e-6)
    # chomp metric
    ck = ka["chomp"]
    n, n_points = ck["n"], ck["n_points"]; m = n_points - 2
    T = np.zeros((n_points, n)); goal = 0.3 * (np.arange(n) + 1)
    for i in range(n_points):
        T[i] = goal * i / (n_points - 1)
    cp = C.POINTER(oracle.Chomp)()
    L.ora_chomp_create(C.byref(cp), m, n, 1, oracle.dp(T[1:].reshape(-1)), n)
    c = cp.contents
    c.dt = ck["dt"]
    Tflat = T.reshape(-1)
    inits = C.cast(c.inits, C.POINTER(oracle.c_double_p)); finals = C.cast(c.finals, C.POINTER(oracle.c_double_p))
    inits[0] = C.cast(Tflat.ctypes.data, oracle.c_double_p)
    finals[0] = C.cast(Tflat.ctypes.data + (n_points - 1) * n * 8, oracle.c_double_p)
    assert L.ora_chomp_init(cp) == 0
    A = np.ctypeslib.as_array(c.A, shape=(m, m)); Ainv = np.ctypeslib.as_array(c.Ainv, shape=(m, m))
    B = np.ctypeslib.as_array(c.B, shape=(m, n)); K = np.ctypeslib.as_array(c.Kvels, shape=(m, m))
    assert np.allclose(A[0, :3], ck["A[0][0..2]"], rtol=1e-12, atol=1e-10)
    assert abs(Ainv[0, 0] - ck["Ainv[0][0]"]) < 1e-12 and abs(Ainv[49, 49] - ck["Ainv[49][49]"]) < 1e-12
    assert K[0, 1] == ck["Kvels[0][1]"] and K[1, 0] == ck["Kvels[1][0]"]
    assert abs(B[98, 0] - ck["B[98][0]"]) < 1e-10 and abs(c.trC - ck["trC"]) < 1e-9
    c.lambda_ = 10.0
    tot, ob, sm = C.c_double(), C.c_double(), C.c_double()
    for _ in range(60):
        L.ora_chomp_iterate(cp, 1, C.byref(tot), C.byref(ob), C.byref(sm))
    assert abs(sm.value - ck["smooth_cost_converged"]) < 1e-9
    assert abs(T[51, 3] - ck["T[50][3]_converged"]) < 1e-9
    # closed form of the inverse of c*tridiag(-1,2,-1): Ainv[i][j] = min(i,j)+1)(m-max(i,j))/((m+1)c)
    i, j = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
    closed = (np.minimum(i, j) + 1) * (m - np.maximum(i, j)) / ((m + 1) * 100.0)
    assert np.allclose(Ainv, closed, rtol=1e-10, atol=0)
    L.ora_chomp_free(cp)


@pytest.mark.parametrize("m,D,free", [(98, 1, 0), (99, 1, 0), (198, 1, 0), (98, 2, 0), (38, 3, 0), (99, 1, 1), (29, 1, 1), (40, 2, 1),
                                      (198, 2, 0), (98, 3, 0), (158, 3, 1), (64, 4, 0), (30, 5, 0), (5, 2, 0)])
def test_product_metric_matches_oracle(oracle, m, D, free):
    """band form + cyclic reduction tables of the product vs the dense A, B, trC, Ainv of the oracle;
    free = the start point is a variable (`start_tsr`: inits[0] == NULL, reference src/orcdchomp_mod.cpp:2572)"""
    Lh = _host(); L = oracle.lib()
    n = 3
    rng = np.random.default_rng(m * 10 + D)
    T = rng.normal(size=(m + 2, n))
    cp = C.POINTER(oracle.Chomp)()
    L.ora_chomp_create(C.byref(cp), m, n, D, oracle.dp(T[1:].reshape(-1)), n)
    c = cp.contents
    dt = 1.0 / (m + 1 - free)
    c.dt = dt
    Tflat = T.reshape(-1)
    inits = C.cast(c.inits, C.POINTER(oracle.c_double_p)); finals = C.cast(c.finals, C.POINTER(oracle.c_double_p))
    inits[0] = C.cast(None if free else Tflat.ctypes.data, oracle.c_double_p)
    finals[0] = C.cast(Tflat.ctypes.data + (m + 1) * n * 8, oracle.c_double_p)
    assert L.ora_chomp_init(cp) == 0
    A = np.ctypeslib.as_array(c.A, shape=(m, m)); Ainv = np.ctypeslib.as_array(c.Ainv, shape=(m, m))
    B = np.ctypeslib.as_array(c.B, shape=(m, n))
    Ah = np.zeros((m, m)); bs = np.zeros(m); bg = np.zeros(m); kap = np.zeros(3)
    rhs = rng.normal(size=(m, n)); sol = np.zeros((m, n))
    dp = oracle.dp
    fn = Lh.orc_host_metric_free_start if free else Lh.orc_host_metric
    assert fn(m, D, dt, dp(Ah), dp(bs), dp(bg), dp(kap), dp(rhs), n, dp(sol)) == 0
    assert np.allclose(Ah, A, rtol=1e-12, atol=1e-9 * np.abs(A).max())
    if free:
        assert not np.any(bs) and kap[0] == 0 and kap[1] == 0
    Bh = np.outer(bs, T[0]) + np.outer(bg, T[-1])
    assert np.allclose(Bh, B, rtol=1e-10, atol=1e-9 * np.abs(B).max())
    trC = 0.5 * (kap[0] * T[0] @ T[0] + 2 * kap[1] * T[0] @ T[-1] + kap[2] * T[-1] @ T[-1])
    assert abs(trC - c.trC) <= 1e-10 * abs(c.trC)
    ref = Ainv @ rhs
    # derivative 2..4: the product applies the band inverse through its rank-D generators (scans), the oracle the dense
    # dgetrf/dgetri-style inverse of the reference; both are within cond(A) eps of the exact solve, so that is the bar
    rank = Lh.orc_host_metric_semisep_rank(m, D, dt, free)
    assert rank == (D if (2 <= D <= 4 and m >= 2 * D + 2) else 0)
    tol = max(1e-9, 50 * np.linalg.cond(A) * 2.0 ** -52)
    assert np.linalg.norm(sol - ref) <= tol * np.linalg.norm(ref), (np.linalg.norm(sol - ref) / np.linalg.norm(ref), tol)
    if rank:
        exact = np.linalg.solve(A.astype(np.longdouble).astype(float), rhs)      # LAPACK's own solve: no explicit inverse
        assert np.linalg.norm(sol - exact) <= tol * np.linalg.norm(exact)
    L.ora_chomp_free(cp)


def test_gsl_stream(oracle):
    """mt19937 against numpy's independent implementation (legacy seeding = init_genrand, which is
    what gsl_rng_set does; seed 0 -> 4357), then oracle == product for the gaussian stream"""
    L = oracle.lib()
    for seed in (0, 1, 4357, 123456789):
        r = oracle.Rng()
        L.ora_rng_set(C.byref(r), seed)
        mine = np.array([L.ora_rng_get(C.byref(r)) for _ in range(2000)], dtype=np.uint64)
        theirs = np.random.RandomState(seed if seed else 4357).randint(0, 2 ** 32, size=2000, dtype=np.uint64)
        assert np.array_equal(mine, theirs)
        r2 = oracle.Rng()
        L.ora_rng_set(C.byref(r2), seed)
        og = np.array([L.ora_ran_gaussian(C.byref(r2), 0.1) for _ in range(1500)])
        ou = L.ora_rng_uniform(C.byref(r2))
        pg = np.zeros(1500); pu = np.zeros(1)
        _host().orc_host_gsl_stream(seed, 0.1, 1500, oracle.dp(pg), oracle.dp(pu))
        assert np.array_equal(og, pg) and ou == pu[0]
        assert abs(og.mean()) < 0.02 and abs(og.std() - 0.1) < 0.01


def test_kin_identities(oracle):
    """pose helpers: compose/invert/compos consistency on 64 random poses (kin.c needs cblas and
    cannot be compiled here; these pin the restatement through algebraic identities)"""
    L = oracle.lib(); dp = oracle.dp
    rng = np.random.default_rng(9)
    for _ in range(64):
        a = rng.normal(size=7); a[3:] /= np.linalg.norm(a[3:])
        b = rng.normal(size=7); b[3:] /= np.linalg.norm(b[3:])
        p = rng.normal(size=3)
        ab = np.zeros(7); ainv = np.zeros(7); ident = np.zeros(7)
        L.ora_kin_pose_compose(dp(a), dp(b), dp(ab))
        L.ora_kin_pose_invert(dp(a), dp(ainv))
        L.ora_kin_pose_compose(dp(a), dp(ainv), dp(ident))
        assert np.allclose(ident, [0, 0, 0, 0, 0, 0, 1], atol=1e-12)
        p1 = np.zeros(3); p2 = np.zeros(3); t = np.zeros(3)
        L.ora_kin_pose_compos(dp(ab), dp(p), dp(p1))
        L.ora_kin_pose_compos(dp(b), dp(p), dp(t)); L.ora_kin_pose_compos(dp(a), dp(t), dp(p2))
        assert np.allclose(p1, p2, atol=1e-12)
        v = np.zeros(3); L.ora_kin_pose_compose_vec(dp(a), dp(p), dp(v))
        assert abs(np.linalg.norm(v) - np.linalg.norm(p)) < 1e-12
        # spatial: the 6x7 pose jacobian maps a quaternion-rate of a pure translation to itself
        J = np.zeros((6, 7)); L.ora_spatial_pose_jac(dp(a), dp(J))
        assert np.array_equal(J[3:, :3], np.eye(3)) and np.array_equal(J[:3, :3], np.zeros((3, 3)))


def test_oracle_fk_jacobian_finite_difference(oracle):
    """the build's own FK/Jacobian (OpenRAVE stand-in): the analytic sphere Jacobian used inside
    sphere_cost_pre equals a finite difference of the sphere positions"""
    prob = common.tabletop_problem(oracle)
    model, base, dofvals, adofs = common.wam_state()
    rob = oracle.OraRobot(model)
    goal = common.wam_goals(1, seed=2)[0]
    p = oracle.default_params(n_points=12, lambda_=100.0, obs_factor=500.0)
    run = oracle.OraRun(rob, base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]], p)
    assert (run.Sa, run.S) == (15, 16)
    assert list(run.sphere_order()) == list(range(1, 16)) + [0]
    G0, c0, P0 = run.eval_obstacle()
    T = run.traj()
    eps = 1e-6
    wp, j = 5, 3
    T[wp, j] += eps
    _, _, P1 = run.eval_obstacle()
    T[wp, j] -= eps
    dP = (P1[wp] - P0[wp]) / eps                    # [Sa][3]
    R, t, ax, an = rob.fk(base, np.r_[T[wp], np.zeros(model.n_dof - 7)])
    link = model.link_names.index("wam4")           # joint j=3 is on link wam4
    for s in range(run.Sa):
        xml = run.sphere_order()[s]
        sl = model.arrays()["sphere_link"][xml]
        affected = sl >= link
        Jcol = np.cross(ax[link], P0[wp, s] - an[link]) if affected else np.zeros(3)
        assert np.allclose(dP[s], Jcol, atol=1e-5)


def test_e2e_config1_regression(oracle):
    """BASELINE configs[0] end to end (tests/golden/make_e2e_golden.py): the oracle reproduces its
    committed trajectories and costs after 1, 10 and 100 iterations"""
    import common
    from or_cdchomp_amd import robots
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "e2e_wam_config1.npz"))
    model, base, dofvals, adofs = common.wam_state()
    prob = common.tabletop_problem(oracle)
    assert list(prob["sizes"]) == list(g["sdf_sizes"]) and np.array_equal(prob["lengths"], g["sdf_lengths"])
    assert np.array_equal(np.asarray(robots.WAM_GOAL), g["goal"])
    rob = oracle.OraRobot(model)
    kw = dict(n_points=101, lambda_=100.0, obs_factor=500.0)
    for n_iter in (1, 10, 100):
        run = oracle.OraRun(rob, base, dofvals, adofs, g["goal"], [prob["sdf"]], [prob["pose"]], oracle.default_params(**kw))
        if n_iter == 1:
            assert np.array_equal(run.traj(), g["seed_traj"])
        st, costs = run.iterate(n_iter)
        assert st == 0
        assert np.allclose(run.traj(), g["traj_%d" % n_iter], rtol=0, atol=1e-12)
        assert np.allclose(costs, g["costs_%d" % n_iter], rtol=1e-12, atol=0)
        run.destroy()


def test_oracle_tsr_constraint_jacobian_and_effect():
    """con_tsr restated (reference src/orcdchomp_mod.cpp:1330-1497, src/libcd/chomp.c:550-600): the constraint
    Jacobian against central differences of the constraint value, and the constraint step pulling a
    violating straight line onto the constraint.  (Parity unpinned: OpenRAVE's end-effector transform and
    Jacobians are the build's own kinematic model.)"""
    from oracle import oracle_py as O
    O.build(ref=False)
    model, _, dofvals, adofs = common.wam_state()
    s2 = np.sqrt(0.5)
    base = [-1.0, 0.0, 1.0, 0.0, s2, 0.0, s2]
    prob = common.tabletop_problem(O)
    rob = O.OraRobot(model)
    rng = np.random.default_rng(3)
    goal = np.array(dofvals[:7]) + 0.4 * rng.uniform(-1, 1, 7)
    run = O.OraRun(rob, base, dofvals, adofs, goal, [prob["sdf"]], [prob["pose"]],
                   O.default_params(n_points=40, lambda_=100.0, obs_factor=200.0))
    ee = model.link_names.index("wam7")
    R, t, _, _ = rob.fk(base, dofvals)
    Bw = [[-1, 1], [-1, 1], [0, 0], [0, 0], [0, 0], [-3, 3]]
    assert run.add_contsr(ee, [0, 0, 0, 0, 0, 0, 1], O.pose_from_dR(t[ee], R[ee]), [0, 0, 0, 0, 0, 0, 1], Bw) == 3
    T = run.traj().copy()
    assert np.allclose(run.eval_contsr(0, T[0])[0], 0.0, atol=1e-12)          # the frame is the start's own
    pt = T[17]
    h, J = run.eval_contsr(0, pt)
    Jfd = np.zeros_like(J)
    for c in range(7):
        d = np.zeros(7); d[c] = 1e-6
        Jfd[:, c] = (run.eval_contsr(0, pt + d)[0] - run.eval_contsr(0, pt - d)[0]) / 2e-6
    assert np.abs(J - Jfd).max() < 1e-8
    before = max(np.abs(run.eval_contsr(0, T[i])[0]).max() for i in range(1, 39))
    st, _ = run.iterate(30)
    T2 = run.traj()
    after = max(np.abs(run.eval_contsr(0, T2[i])[0]).max() for i in range(1, 39))
    assert st == 0 and before > 0.1 and after < 1e-5
    run.destroy()


def test_oracle_dgesv_leaves_the_right_hand_side_alone_when_singular(oracle):
    """LAPACKE_dgesv never calls dgetrs when dgetrf reports a zero pivot: chomp.c:582-599 then pushes the ORIGINAL h
    back through A^-1 J^T ("constraint inversion error!").  The oracle's LU does the same; a regular system is solved."""
    import ctypes as C
    L = oracle.lib()
    L.ora_dgesv_one.argtypes = [C.c_int, oracle.c_double_p, oracle.c_int_p, oracle.c_double_p]
    rng = np.random.default_rng(5)
    for n in (1, 3, 8, 17):
        A = rng.normal(size=(n, n)); b = rng.normal(size=n)
        A2, b2 = A.copy(), b.copy(); piv = np.zeros(n, dtype=np.int32)
        assert L.ora_dgesv_one(n, oracle.dp(A2), oracle.ip(piv), oracle.dp(b2)) == 0
        assert np.allclose(b2, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)
    # a constraint given twice: two identical rows and columns
    J = rng.normal(size=(3, 7)); J2 = np.vstack([J, J])
    S = J2 @ J2.T
    h = rng.normal(size=6); h2 = h.copy(); piv = np.zeros(6, dtype=np.int32)
    info = L.ora_dgesv_one(6, oracle.dp(np.ascontiguousarray(S)), oracle.ip(piv), oracle.dp(h2))
    if info > 0:                                       # (an exact zero pivot is a matter of rounding)
        assert np.array_equal(h2, h)
    Z = np.zeros((4, 4)); Z[0, 0] = 1.0
    h = rng.normal(size=4); h2 = h.copy(); piv = np.zeros(4, dtype=np.int32)
    assert L.ora_dgesv_one(4, oracle.dp(Z), oracle.ip(piv), oracle.dp(h2)) == 2
    assert np.array_equal(h2, h)
