"""-m gpu: what the C ABI (include/orcdchomp_amd.h) does with a caller's mistakes.  The reference reports errors as
exceptions through OpenRAVE (SURVEY.md 8b "Validation / error convention"); across a C boundary they are return codes
(1 + orc_last_error), and a null pointer or a malformed robot description must come back as one too -- never as a
fault inside the library, which would take the caller's process (a planner, an OpenRAVE plugin host) with it."""
import ctypes as C

import numpy as np
import pytest

import common
from or_cdchomp_amd import _capi, robots, scenes

pytestmark = pytest.mark.gpu


def _err(mod):
    return mod._lib.orc_last_error(mod._h).decode()


def _desc(model):
    a = model.arrays()
    keep = [np.ascontiguousarray(a[k]) for k in ("parent", "pose_parent_joint", "joint_type", "axis", "dof_index", "limit_lower",
                                                  "limit_upper", "sphere_link", "sphere_pos", "sphere_radius")]
    ip = lambda x: x.ctypes.data_as(C.POINTER(C.c_int))
    dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    d = _capi.RobotDesc()
    d.n_links = a["n_links"]; d.parent = ip(keep[0]); d.pose_parent_joint = dp(keep[1]); d.joint_type = ip(keep[2])
    d.axis = dp(keep[3]); d.dof_index = ip(keep[4]); d.n_dof = a["n_dof"]; d.limit_lower = dp(keep[5]); d.limit_upper = dp(keep[6])
    d.n_spheres = a["n_spheres"]; d.sphere_link = ip(keep[7]); d.sphere_pos = dp(keep[8]); d.sphere_radius = dp(keep[9])
    return d, keep


def test_null_pointers_and_bad_handles_are_error_codes():
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(0)
    L, h = mod._lib, mod._h
    model = robots.wam7()
    d, keep = _desc(model)
    # no module: code 2, nothing touched
    assert L.orc_batch_destroy(None, 1) == 2 and L.orc_send_command(None, b"destroy run 1", None, 0) == 2
    # robot registration
    assert L.orc_env_add_robot(h, None, C.byref(d)) == 1 and "null argument: name" in _err(mod)
    assert L.orc_env_add_robot(h, b"r", None) == 1 and "null argument" in _err(mod)
    for field, value, msg in (("joint_type", 7, "joint type"), ("dof_index", 99, "dof index"), ("sphere_radius", -1.0, "sphere radius"),
                              ("sphere_radius", float("nan"), "sphere radius"), ("axis", 0.0, "joint axis"), ("sphere_link", 99, "does not exist"),
                              ("parent", 5, "topological order")):
        idx = {"joint_type": 2, "dof_index": 2, "sphere_radius": 3, "axis": None, "sphere_link": 1, "parent": 2}[field]
        arr = keep[["parent", "pose_parent_joint", "joint_type", "axis", "dof_index", "limit_lower", "limit_upper", "sphere_link",
                    "sphere_pos", "sphere_radius"].index(field)]
        saved = arr.copy()
        if field == "axis":
            arr[1] = 0.0
        else:
            arr[idx] = value
        assert L.orc_env_add_robot(h, b"bad", C.byref(d)) == 1, field
        assert msg in _err(mod), (field, _err(mod))
        arr[...] = saved
    d.n_links = 0
    assert L.orc_env_add_robot(h, b"bad", C.byref(d)) == 1 and "counts" in _err(mod)
    d.n_links = len(model.link_names)
    assert L.orc_env_add_robot(h, b"arm", C.byref(d)) == 0
    assert L.orc_env_add_robot(h, b"arm", C.byref(d)) == 1 and "already exists" in _err(mod)
    # setters: unknown names, null arrays, wrong lengths
    pose = (C.c_double * 7)(0, 0, 0, 0, 0, 0, 1)
    assert L.orc_robot_set_transform(h, b"nobody", pose) == 1 and "Could not find robot" in _err(mod)
    assert L.orc_robot_set_transform(h, None, pose) == 1 and L.orc_robot_set_transform(h, b"arm", None) == 1
    assert L.orc_robot_set_dof_values(h, b"arm", None, model.n_dof) == 1 and "null argument" in _err(mod)
    assert L.orc_robot_set_dof_values(h, b"arm", None, 3) == 1 and "wrong number" in _err(mod)
    assert L.orc_robot_set_active_dofs(h, b"arm", None, 2) == 1
    bad = (C.c_int * 2)(0, 99)
    assert L.orc_robot_set_active_dofs(h, b"arm", bad, 2) == 1 and "bad dof index" in _err(mod)
    assert L.orc_robot_set_velocity_limits(h, b"arm", None, model.n_dof) == 1
    assert L.orc_robot_set_link_names(h, b"arm", None, len(model.link_names)) == 1
    assert L.orc_robot_add_manipulator(h, b"arm", None, 1, None) == 1
    assert L.orc_robot_add_manipulator(h, b"arm", b"m", 999, None) == 1 and "out of range" in _err(mod)
    assert L.orc_robot_set_active_manipulator(h, b"arm", None) == 1
    assert L.orc_robot_set_adjacent_links(h, b"arm", None, 2) == 1
    assert L.orc_set_workgroup_threads(h, 100) == 1 and L.orc_set_workgroups_per_cu(h, 5) == 1
    # scene
    assert L.orc_env_add_kinbody_boxes(h, None, 0, None, None) == 1
    assert L.orc_env_add_kinbody_boxes(h, b"box", 2, None, None) == 1 and "null argument" in _err(mod)
    assert L.orc_env_add_kinbody_boxes(h, b"box", -1, None, None) == 1
    scenes.add_tabletop(mod)
    # bodies the robot holds (orc_kinbody_set_spheres, orc_robot_grab / _release, orc_body_get_transform, orc_robot_set_self_check)
    one = (C.c_double * 3)(0, 0, 0); rad = (C.c_double * 1)(0.05)
    assert L.orc_kinbody_set_spheres(h, None, 1, one, rad) == 1 and L.orc_kinbody_set_spheres(h, b"ghost", 1, one, rad) == 1
    assert "Could not find kinbody" in _err(mod)
    assert L.orc_kinbody_set_spheres(h, b"mug", 1, None, rad) == 1 and L.orc_kinbody_set_spheres(h, b"mug", 1, one, None) == 1
    assert L.orc_kinbody_set_spheres(h, b"mug", -1, one, rad) == 1 and "bad number of spheres" in _err(mod)
    assert L.orc_kinbody_set_spheres(h, b"mug", 0, None, None) == 0 and L.orc_kinbody_set_spheres(h, b"mug", 1, one, rad) == 0
    assert L.orc_robot_grab(h, None, b"mug", 1) == 1 and L.orc_robot_grab(h, b"arm", None, 1) == 1
    assert L.orc_robot_grab(h, b"nobody", b"mug", 1) == 1 and "Could not find robot" in _err(mod)
    assert L.orc_robot_grab(h, b"arm", b"ghost", 1) == 1 and "Could not find kinbody" in _err(mod)
    assert L.orc_robot_grab(h, b"arm", b"mug", -1) == 1 and L.orc_robot_grab(h, b"arm", b"mug", 999) == 1 and "out of range" in _err(mod)
    assert L.orc_robot_release(h, b"arm", b"mug") == 1 and "not grabbing" in _err(mod)
    assert L.orc_robot_grab(h, b"arm", b"mug", 3) == 0 and L.orc_robot_grab(h, b"arm", b"mug", 3) == 1 and "already grabbed" in _err(mod)
    got = (C.c_double * 7)()
    assert L.orc_body_get_transform(h, b"mug", None) == 1 and L.orc_body_get_transform(h, b"ghost", got) == 1 and L.orc_body_get_transform(h, None, got) == 1
    assert L.orc_body_get_transform(h, b"mug", got) == 0 and L.orc_body_get_transform(h, b"arm", got) == 0
    assert L.orc_robot_release(h, b"arm", None) == 1 and L.orc_robot_release(h, b"arm", b"mug") == 0
    assert L.orc_robot_release_all(h, None) == 1 and L.orc_robot_release_all(h, b"nobody") == 1 and L.orc_robot_release_all(h, b"arm") == 0
    assert L.orc_robot_set_self_check(h, None, 0) == 1 and L.orc_robot_set_self_check(h, b"nobody", 0) == 1 and L.orc_robot_set_self_check(h, b"arm", 1) == 0
    sizes = (C.c_int * 3)(4, 4, 1); lengths = (C.c_double * 3)(0.1, 0.1, 0.1)
    data = (C.c_double * 64)()
    assert L.orc_scene_add_sdf(h, b"table", sizes, lengths, pose, data) == 1 and "at least 2 cells" in _err(mod)
    assert L.orc_scene_add_sdf(h, b"table", None, lengths, pose, data) == 1 and L.orc_scene_add_sdf(h, b"table", sizes, lengths, pose, None) == 1
    assert L.orc_scene_add_sdf(h, b"ghost", sizes, lengths, pose, data) == 1 and "Could not find kinbody" in _err(mod)
    assert L.orc_scene_get_sdf(h, b"table", sizes, lengths, pose, None, 0) == 1 and "No sdf" in _err(mod)
    # batches: the create command's own checks (src/orcdchomp_mod.cpp:2091-2097), then handles and buffers
    p = _capi.BatchParams(); L.orc_batch_params_default(C.byref(p)); L.orc_batch_params_default(None)
    bid = C.c_int(0)
    goals = np.zeros((2, model.n_dof)); gp = goals.ctypes.data_as(C.POINTER(C.c_double))
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 1
    assert "No signed distance fields have yet been computed!" in _err(mod)
    mod.SendCommand("computedistancefield kinbody table")
    assert L.orc_batch_create(h, b"arm", None, 2, None, gp, None, None, C.byref(bid)) == 1
    assert L.orc_batch_create(h, None, C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 1
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, None) == 1
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, None, None, None, C.byref(bid)) == 1 and "Did not pass either adofgoal or starttraj!" in _err(mod)
    assert L.orc_batch_create(h, b"arm", C.byref(p), 0, None, gp, None, None, C.byref(bid)) == 1
    p.lambda_ = 0.001
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 1 and "lambda must be >=0.01!" in _err(mod)
    p.lambda_ = 100.0; p.n_points = 2
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 1 and "n_points must be >=3!" in _err(mod)
    p.n_points = 12; p.floating_base = 1
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 1 and "Passed floating_base with no basegoal!" in _err(mod)
    p.floating_base = 0
    assert L.orc_batch_create(h, b"arm", C.byref(p), 2, None, gp, None, None, C.byref(bid)) == 0, _err(mod)
    b = bid.value
    for call in (lambda: L.orc_batch_iterate(h, b + 17, 1, None, None), lambda: L.orc_batch_gettraj(h, b + 17, gp, 10),
                 lambda: L.orc_batch_destroy(h, b + 17), lambda: L.orc_batch_dims(h, -3, None, None, None)):
        assert call() == 1 and "you must pass a created run!" in _err(mod)
    assert L.orc_batch_iterate(h, b, -1, None, None) == 1 and "n_iter must be >=0!" in _err(mod)
    assert L.orc_batch_iterate(h, b, 2, None, None) == 0
    small = np.zeros(5)
    assert L.orc_batch_gettraj(h, b, small.ctypes.data_as(C.POINTER(C.c_double)), small.size) == 1 and "buffer too small" in _err(mod)
    assert L.orc_batch_gettraj(h, b, None, 10**9) == 1
    assert L.orc_batch_get_trace(h, b, None, 10**9) == 1 and L.orc_batch_get_state(h, b, None, gp, 10**9) == 1
    assert L.orc_batch_get_state(h, b, b"nonsense", np.zeros(2 * 10 * model.n_dof).ctypes.data_as(C.POINTER(C.c_double)), 10**9) == 1
    assert L.orc_batch_set_traj(h, b, None, 2 * 12 * model.n_dof) == 1 and L.orc_batch_set_traj(h, b, gp, 5) == 1
    assert L.orc_batch_set_noise(h, b, None, 3) == 1
    assert L.orc_batch_collision_verdict(h, b, None, None, None, None, None) == 1
    # and after all that the module still works
    costs = np.zeros((2, 3)); status = np.zeros(2, dtype=np.int32)
    assert L.orc_batch_iterate(h, b, 3, costs.ctypes.data_as(C.POINTER(C.c_double)), status.ctypes.data_as(C.POINTER(C.c_int))) == 0
    assert np.all(np.isfinite(costs)) and _err(mod) == ""
    assert L.orc_batch_destroy(h, b) == 0 and L.orc_batch_destroy(h, b) == 1
