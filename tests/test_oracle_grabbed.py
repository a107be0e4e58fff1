"""CPU: the oracle's restatement of create's sphere collection for a robot that holds bodies
(reference src/orcdchomp_mod.cpp:2148-2300), checked against an independent numpy reading of the same lines."""
import numpy as np
import pytest

import common
from or_cdchomp_amd import robots

KW = dict(n_points=20, lambda_=100.0, obs_factor=500.0)


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)], [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                     [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])


@pytest.fixture(scope="module")
def wam(oracle):
    model = robots.wam7()
    base = [-1.0, 0.0, 1.0, 0.0, np.sqrt(0.5), 0.0, np.sqrt(0.5)]
    q = np.zeros(model.n_dof); q[:7] = robots.WAM_START
    return dict(model=model, base=base, q=q, prob=common.tabletop_problem(oracle))


def _run(oracle, wam, grabbed, adofs=range(7), goal=None, **kw):
    rob = oracle.OraRobot(wam["model"], grabbed=grabbed)
    adofs = list(adofs)
    goal = np.zeros(len(adofs)) if goal is None else goal
    params = dict(KW); params.update(kw)
    run = oracle.OraRun(rob, wam["base"], wam["q"], adofs, goal, [wam["prob"]["sdf"]], [wam["prob"]["pose"]],
                        oracle.default_params(**params))
    run._rob = rob
    return run


def test_list_order_is_the_head_insertion_of_create(oracle, wam):
    """robot first, then the grabbed bodies, every sphere to the HEAD of the list (mod.cpp:2273-2290): the last body's
    spheres come first, the robot's last; a body's own spheres keep their XML order (the kdata list is reversed once, the
    head insertion once more); inactive spheres likewise behind the active ones"""
    m = wam["model"]
    hand, fore, base_link = m.link_names.index("handbase"), m.link_names.index("wam4"), m.link_names.index("wam0")
    ident = [0.5, 0.2, 0.9, 0, 0, 0, 1]
    A = (hand, ident, [[0, 0, 0], [0.1, 0, 0]], [0.03, 0.04])                   # XML indices 16, 17: active
    B = (base_link, ident, [[0, 0, 0.5]], [0.05])                              # 18: inactive (the base link)
    Cc = (fore, ident, [[0, 0, 0], [0, 0.1, 0], [0, 0.2, 0]], [0.02, 0.02, 0.02])   # 19, 20, 21: active
    run = _run(oracle, wam, [A, B, Cc])
    assert (run.S, run.Sa) == (22, 20)
    order = list(run.sphere_order())
    assert order == [19, 20, 21, 16, 17] + list(range(1, 16)) + [18, 0], order
    run.destroy()
    # only the first four arm joints active: the hand's body is still active (its link is moved by them)
    run = _run(oracle, wam, [A], adofs=range(4))
    assert run.Sa == 17 and list(run.sphere_order())[:2] == [16, 17]
    run.destroy()


def test_held_sphere_sits_where_the_body_is(oracle, wam):
    """pos_wrt_link = T_w_rlink^-1 o T_w_klink o pos (mod.cpp:2200-2208): at create the sphere is at T_w_klink o pos in the
    world, and it follows its link afterwards"""
    m = wam["model"]
    hand = m.link_names.index("handbase")
    kpose = [-0.3, 0.4, 1.1] + robots.quat_from_axis_angle((1, 2, -1), 0.9)
    pos = np.array([[0.02, -0.01, 0.05], [0.1, 0.0, 0.0]])
    goal = np.array(robots.WAM_GOAL)
    run = _run(oracle, wam, [(hand, kpose, pos, [0.03, 0.03])], goal=goal)
    _, _, P = run.eval_obstacle()                                  # sphere_poss_all [n_points][S_a][3], list order
    order = list(run.sphere_order())
    want0 = (_rot(kpose[3:]) @ pos.T).T + np.asarray(kpose[:3])
    for k, xml in enumerate((16, 17)):
        assert np.allclose(P[0, order.index(xml)], want0[k], rtol=0, atol=1e-14)
    # at the goal: carried by the hand frame
    R0, t0 = m.link_frames(wam["base"], wam["q"])
    qg = wam["q"].copy(); qg[:7] = goal
    R1, t1 = m.link_frames(wam["base"], qg)
    for k, xml in enumerate((16, 17)):
        local = R0[hand].T @ (want0[k] - t0[hand])
        assert np.allclose(P[-1, order.index(xml)], R1[hand] @ local + t1[hand], rtol=0, atol=1e-13)
    run.destroy()


def test_a_body_without_spheres_is_an_error(oracle, wam):
    m = wam["model"]
    with pytest.raises(RuntimeError, match="no spheres! kinbody does not have a <orcdchomp> tag defined\\?"):
        _run(oracle, wam, [(3, [0, 0, 0, 0, 0, 0, 1], np.zeros((0, 3)), [])])
    import copy
    naked = copy.deepcopy(m); naked.spheres = []
    with pytest.raises(RuntimeError, match="no spheres!"):
        _run(oracle, dict(wam, model=naked), [])


def test_recheck_sees_the_held_spheres(oracle, wam):
    """the re-check walks the run's spheres, the held ones included (mod.cpp:2992-2996): a sphere held half a metre below the
    hand dips into the table where the arm itself stays clear"""
    m = wam["model"]
    hand = m.link_names.index("handbase")
    R, t = m.link_frames(wam["base"], wam["q"])
    found = 0
    for goal in common.wam_goals(60, seed=3):
        bare = _run(oracle, wam, [], goal=goal, n_points=40)
        none = bare.collision_recheck(np.ones(7))
        bare.destroy()
        if none["collides"]:
            continue
        for reach in (0.3, 0.5, 0.7):
            kpose = list(t[hand] + R[hand] @ np.array([0, 0, reach])) + [0, 0, 0, 1]
            run = _run(oracle, wam, [(hand, kpose, [[0, 0, 0]], [0.05])], goal=goal, n_points=40)
            hit = run.collision_recheck(np.ones(7))
            run.destroy()
            if hit["collides"]:
                assert hit["sphere"] == 16 and hit["field"] == 0 and hit["depth"] > 0
                found += 1
                break
    assert found >= 1


def test_spheres_geometry_group_order(oracle, wam):
    """OpenRAVE >= 0.9: a body's spheres are its <orcdchomp> kdata spheres AND the GT_Sphere geometries of the "spheres" group of its
    links (mod.cpp:2214-2259), appended after them and then pushed onto the head of the run's list: the kdata spheres end in XML
    order (reversed twice), the group's spheres reversed.  The binding (INTEGRATION.md "Spheres from the spheres geometry group")
    hands a body's spheres over as ONE array -- the group's last (link, geometry) first, then the kdata spheres in XML order --
    and that array is the run's order: the literal restatement of the reference's list handling says the same, and the C
    oracle's list (which takes such an array as "XML order") agrees with it."""
    m = wam["model"]
    # ids: the robot's 16 kdata spheres 0..15 in XML order, four group spheres 100..103 on links in index order
    inactive_links = {m.link_names.index("wam0")}
    link_of = {i: m.link_names.index(s[0]) for i, s in enumerate(m.spheres)}
    group = [100, 101, 102, 103]
    group_link = {100: m.link_names.index("wam0"), 101: m.link_names.index("wam2"), 102: m.link_names.index("wam2"), 103: m.link_names.index("handbase")}
    link_of.update(group_link)
    active = {i for i, l in link_of.items() if l not in inactive_links}
    robot = dict(xml=list(range(16)), group=group, active=active)
    held = dict(xml=[200, 201], group=[210, 211, 212], active={200, 201, 210, 211, 212})
    ref = oracle.reference_sphere_list([robot, held])
    # the binding's rule, body by body: reversed(group) + xml; the last grabbed body first, the robot last; inactive ones behind
    def binding(body):
        return list(reversed(body["group"])) + list(body["xml"])
    want_active = [i for i in binding(held) if i in held["active"]] + [i for i in binding(robot) if i in robot["active"]]
    want_inactive = [i for i in binding(held) if i not in held["active"]] + [i for i in binding(robot) if i not in robot["active"]]
    assert ref == want_active + want_inactive, (ref, want_active, want_inactive)
    assert ref[:5] == [212, 211, 210, 200, 201] and ref[5:8] == [103, 102, 101] and ref[-2:] == [100, 0]
    # a body without a group: plain XML order (what the oracle and the product were tested with all along)
    assert oracle.reference_sphere_list([dict(xml=[0, 1, 2], active={0, 1, 2})]) == [0, 1, 2]
    with pytest.raises(RuntimeError, match="no spheres"):
        oracle.reference_sphere_list([dict(xml=[], group=[], active=set())])
    # the C oracle with the robot's array in binding order: its list is that array's active part, then the inactive part
    import copy
    model = copy.deepcopy(m)
    grp = [("wam0", [0.0, 0.0, 0.1], 0.05), ("wam2", [0.0, 0.0, 0.05], 0.04), ("wam2", [0.0, 0.05, 0.0], 0.03), ("handbase", [0.0, 0.0, 0.02], 0.02)]
    model.spheres = list(reversed(grp)) + list(m.spheres)                 # what collect_spheres of INTEGRATION.md hands over
    run = oracle.OraRun(oracle.OraRobot(model), wam["base"], wam["q"], list(range(7)), np.zeros(7), [wam["prob"]["sdf"]], [wam["prob"]["pose"]],
                        oracle.default_params(**KW))
    order = list(run.sphere_order())                                      # indices into model.spheres
    ids = [103, 102, 101, 100] + list(range(16))                          # the ids of that array's entries
    assert [ids[k] for k in order] == [i for i in ref if i < 200], ([ids[k] for k in order], ref)
    run.destroy()


def test_a_held_body_does_not_change_the_robots_own_link_pairs(oracle, wam):
    """the self-collision leg of the re-check (mod.cpp:2998-2999): which link pairs of the ROBOT are tested comes from the robot's own
    spheres; a held body is left out against its holder link and against the links it touches in the configuration of create,
    and only it -- a fat body that reaches the forearm does not switch off the hand-forearm pairs of the robot's own spheres"""
    m = wam["model"]
    hand, fore = m.link_names.index("handbase"), m.link_names.index("wam4")
    bare = _run(oracle, wam, [])
    ex0 = bare.self_excluded()
    n_own = bare.S
    bare.destroy()
    # a body in the hand whose second sphere is put onto the forearm's spheres (it touches wam4 at create)
    R, t, _, _ = oracle.OraRobot(m).fk(wam["base"], wam["q"])
    fore_sphere = next(i for i, s in enumerate(m.spheres) if m.link_names.index(s[0]) == fore)
    p_fore = R[fore] @ np.asarray(m.spheres[fore_sphere][1]) + t[fore]
    pose = list(t[hand]) + [0, 0, 0, 1]
    held = (hand, pose, [[0, 0, 0.1], list(p_fore - t[hand])], [0.04, 0.05])
    run = _run(oracle, wam, [held])
    ex = run.self_excluded()
    run.destroy()
    assert ex.shape == (n_own + 2, n_own + 2)
    assert np.array_equal(ex[:n_own, :n_own], ex0)                              # the robot's own pairs are what they were
    assert ex[n_own, n_own + 1] == 1                                             # one rigid body
    link = np.array([m.link_names.index(s[0]) for s in m.spheres])
    for a in (n_own, n_own + 1):
        assert ex[a, :n_own][link == hand].all()                                 # never against the link that holds it
        assert ex[a, :n_own][link == fore].all()                                 # nor against the link it touched at create (either sphere: the BODY touched it)
    far = [i for i in range(n_own) if link[i] not in (hand, fore) and not ex[n_own + 1, i]]
    assert len(far) > 0                                                          # but against the links it did not touch
    # the same body held clear of everything: tested against every link but its holder
    clear = (hand, list(t[hand] + R[hand] @ np.array([0.0, 0.0, 0.45])) + [0, 0, 0, 1], [[0, 0, 0]], [0.03])
    run = _run(oracle, wam, [clear])
    ex = run.self_excluded()
    run.destroy()
    assert np.array_equal(ex[n_own, :n_own] == 1, link == hand)


def test_grab_contacts_are_taken_at_the_grab_not_at_create(oracle, wam):
    """OpenRAVE records what a body touches at the moment of RobotBase::Grab and CheckSelfCollision leaves exactly that out
    (src/orcdchomp_mod.cpp:2998-2999 calls it).  A body grabbed with the wrist straight touches the wrist link only; when the
    run is created with the wrist bent so far that the body meets the forearm, the forearm pair counts.  The same body grabbed
    in that bent state is left out against the forearm (round-5 advisor: the exclusions used to be taken at create)."""
    m = wam["model"]
    hand, fore, wrist = m.link_names.index("handbase"), m.link_names.index("wam4"), m.link_names.index("wam6")
    q_grab = wam["q"].copy(); q_grab[5] = 0.0
    q_create = wam["q"].copy(); q_create[5] = 1.5
    pos = [[0.0, 0.0, 0.10], [0.15, 0.0, 0.03]]; rad = [0.04, 0.07]
    # the body's own frame is the hand's (it rides with it): its world pose at create
    R, t = m.link_frames(wam["base"], q_create)
    pose_create = list(oracle.pose_from_dR(t[hand], R[hand]))
    link_of = np.array([m.link_names.index(s[0]) for s in m.spheres])

    def world(q):
        Rq, tq = m.link_frames(wam["base"], q)
        own = [(link_of[i], Rq[link_of[i]] @ np.asarray(s[1]) + tq[link_of[i]], s[2]) for i, s in enumerate(m.spheres)]
        held = [(Rq[hand] @ np.asarray(p) + tq[hand], r) for p, r in zip(pos, rad)]
        return own, held

    def touched(q):
        own, held = world(q)
        return {int(l) for l, c, r in own for p, rr in held if np.linalg.norm(p - c) - (r + rr) < 0.0}
    assert fore not in touched(q_grab) and fore in touched(q_create) and wrist in touched(q_grab)      # the scene is what the test says

    def excl(grabbed):
        rob = oracle.OraRobot(m, grabbed=grabbed)
        run = oracle.OraRun(rob, wam["base"], q_create, list(range(7)), np.zeros(7), [wam["prob"]["sdf"]], [wam["prob"]["pose"]],
                            oracle.default_params(**KW))
        ex = run.self_excluded().copy()
        run.destroy()
        return ex
    at_grab = excl([(hand, pose_create, pos, rad, wam["base"], q_grab)])
    at_create = excl([(hand, pose_create, pos, rad)])
    n = len(m.spheres)
    for ex, fore_out in ((at_grab, False), (at_create, True)):
        assert ex.shape == (n + 2, n + 2) and (ex == ex.T).all()
        assert ex[n:, :n][:, link_of == hand].all() if (link_of == hand).any() else True      # the grabbing link, always
        assert ex[n:, :n][:, link_of == wrist].all()                                           # touched at either moment
        assert ex[n:, :n][:, link_of == fore].all() == fore_out
        assert ex[n, n + 1] and ex[n + 1, n]                                                   # one rigid body
    assert not at_grab[n:, :n][:, link_of == fore].any()
    # an independent reading of the rule: a held sphere against a robot sphere is left out iff the BODY touched that sphere's link then
    for q, ex in ((q_grab, at_grab), (q_create, at_create)):
        tl = touched(q) | {hand}
        want = np.array([[link_of[i] in tl for i in range(n)]] * 2)
        assert np.array_equal(ex[n:, :n].astype(bool), want)


def test_two_held_bodies_touch_at_the_later_grab(oracle, wam):
    """two bodies on two links: whether they are left out against each other is decided where they were when the SECOND was grabbed"""
    m = wam["model"]
    hand, fore = m.link_names.index("handbase"), m.link_names.index("wam4")
    q_apart = wam["q"].copy(); q_apart[5] = 0.0
    q_touch = wam["q"].copy(); q_touch[5] = 1.5
    R, t = m.link_frames(wam["base"], q_touch)
    pose_hand = list(oracle.pose_from_dR(t[hand], R[hand])); pose_fore = list(oracle.pose_from_dR(t[fore], R[fore]))
    A = [[0.15, 0.0, 0.03]], [0.07]
    # body B sits on the forearm where A ends up when the wrist is bent
    pB = np.linalg.inv(R[fore]) @ (R[hand] @ np.asarray(A[0][0]) + t[hand] - t[fore])
    B = [list(pB)], [0.05]
    n = len(m.spheres)

    def excl(state_second):
        second = (fore, pose_fore, B[0], B[1]) + ((wam["base"], state_second) if state_second is not None else ())
        rob = oracle.OraRobot(m, grabbed=[(hand, pose_hand, A[0], A[1], wam["base"], q_apart), second])
        run = oracle.OraRun(rob, wam["base"], q_touch, list(range(7)), np.zeros(7), [wam["prob"]["sdf"]], [wam["prob"]["pose"]],
                            oracle.default_params(**KW))
        ex = run.self_excluded().copy()
        run.destroy()
        return ex
    assert not excl(q_apart)[n, n + 1]           # apart when B was grabbed: the pair counts, although they touch at create
    assert excl(q_touch)[n, n + 1]               # touching when B was grabbed: left out
    assert excl(None)[n, n + 1]                  # (no state given: grabbed in the state of create)
