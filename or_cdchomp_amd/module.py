"""Host-side Python mirror of the orcdchomp module object.

``Module.SendCommand`` takes the same command strings as the reference's OpenRAVE
module (/root/reference src/orcdchomp_mod.h:58-66) and returns the same text;
errors surface as ``RuntimeError`` carrying the reference's exception message
(openravepy turns openrave_exception into a Python exception the same way).
Everything is executed by liborcdchomp_amd.so through its C ABI.
"""
import ctypes as C

import numpy as np

from . import _capi


def _dp(a):
    return a.ctypes.data_as(_capi.c_double_p)


def _ip(a):
    return a.ctypes.data_as(_capi.c_int_p)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Module:
    def __init__(self, device=0):
        """device: one HIP ordinal, or a list of them (batches are then sharded over the list inside
        this process, host-side gather, no collective)."""
        self._lib = _capi.lib()
        if isinstance(device, (list, tuple)):
            devs = np.ascontiguousarray(device, dtype=np.int32)
            self._h = self._lib.orc_module_new_multi(_ip(devs), len(devs))
        else:
            self._h = self._lib.orc_module_new(int(device))
        if not self._h:
            raise RuntimeError(self._lib.orc_last_error(None).decode())
        self._keep = []

    def close(self):
        if getattr(self, "_h", None):
            self._lib.orc_module_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.orc_last_error(self._h).decode())

    # ---- the reference's surface ------------------------------------------------
    def SendCommand(self, cmd, releasegil=False):
        buf = C.create_string_buffer(4096)
        self._check(self._lib.orc_send_command(self._h, cmd.encode(), buf, len(buf)))
        size = self._lib.orc_last_reply_size(self._h)
        if size >= len(buf):
            big = C.create_string_buffer(size + 1)
            self._lib.orc_last_reply(self._h, big, len(big))
            return big.value.decode()
        return buf.value.decode()

    def set_stream(self, hip_stream):
        self._check(self._lib.orc_set_stream(self._h, C.c_void_p(hip_stream)))

    def set_workgroup_threads(self, threads):
        """0 (default plan), 192 or 256 threads per workgroup for the batches created from now on"""
        self._check(self._lib.orc_set_workgroup_threads(self._h, int(threads)))

    def set_workgroups_per_cu(self, workgroups):
        """0 (default budget) or 4: four 256-thread workgroups per CU at 128 registers where a kernel is built for it"""
        self._check(self._lib.orc_set_workgroups_per_cu(self._h, int(workgroups)))

    def set_num_streams(self, n):
        self._check(self._lib.orc_set_num_streams(self._h, int(n)))

    # ---- environment stand-ins -----------------------------------------------------
    def add_robot(self, model, transform=None, dof_values=None, active_dofs=None):
        a = model.arrays()
        d = _capi.RobotDesc()
        d.n_links = a["n_links"]
        d.parent = _ip(a["parent"]); d.pose_parent_joint = _dp(a["pose_parent_joint"])
        d.joint_type = _ip(a["joint_type"]); d.axis = _dp(a["axis"]); d.dof_index = _ip(a["dof_index"])
        d.n_dof = a["n_dof"]
        d.limit_lower = _dp(a["limit_lower"]); d.limit_upper = _dp(a["limit_upper"])
        d.n_spheres = a["n_spheres"]
        d.sphere_link = _ip(a["sphere_link"]); d.sphere_pos = _dp(a["sphere_pos"])
        d.sphere_radius = _dp(a["sphere_radius"])
        self._check(self._lib.orc_env_add_robot(self._h, model.name.encode(), C.byref(d)))
        names = (C.c_char_p * len(model.link_names))(*[nm.encode() for nm in model.link_names])
        self._check(self._lib.orc_robot_set_link_names(self._h, model.name.encode(), names, len(model.link_names)))
        for mname, link, tool in getattr(model, "manipulators", []):
            self.add_manipulator(model.name, mname, model.link_names.index(link), tool)
        adj = getattr(model, "adjacent", [])
        if adj:
            pairs = np.ascontiguousarray([[model.link_names.index(a), model.link_names.index(b)] for a, b in adj], dtype=np.int32)
            self._check(self._lib.orc_robot_set_adjacent_links(self._h, model.name.encode(), _ip(pairs), len(adj)))
        if transform is not None:
            self.set_robot_transform(model.name, transform)
        if dof_values is not None:
            self.set_dof_values(model.name, dof_values)
        if active_dofs is not None:
            self.set_active_dofs(model.name, active_dofs)

    def add_manipulator(self, robot, name, ee_link, tool_pose=(0, 0, 0, 0, 0, 0, 1)):
        """a manipulator: end-effector link index + local tool transform (GetEndEffectorTransform = link o tool);
        the first one added is the active manipulator"""
        self._check(self._lib.orc_robot_add_manipulator(self._h, robot.encode(), name.encode(), int(ee_link), _dp(_f64(tool_pose))))

    def set_active_manipulator(self, robot, name):
        self._check(self._lib.orc_robot_set_active_manipulator(self._h, robot.encode(), name.encode()))

    def set_self_check(self, robot, enabled=True):
        """the sphere-pair stand-in for CheckSelfCollision in gettraj's re-check, per robot (default on)"""
        self._check(self._lib.orc_robot_set_self_check(self._h, robot.encode(), 1 if enabled else 0))

    def set_robot_transform(self, name, pose):
        self._check(self._lib.orc_robot_set_transform(self._h, name.encode(), _dp(_f64(pose))))

    def set_dof_values(self, name, values):
        v = _f64(values)
        self._check(self._lib.orc_robot_set_dof_values(self._h, name.encode(), _dp(v), len(v)))

    def set_active_dofs(self, name, indices):
        idx = np.ascontiguousarray(indices, dtype=np.int32)
        self._check(self._lib.orc_robot_set_active_dofs(self._h, name.encode(), _ip(idx), len(idx)))

    def set_velocity_limits(self, name, limits):
        v = _f64(limits)
        self._check(self._lib.orc_robot_set_velocity_limits(self._h, name.encode(), _dp(v), len(v)))

    def last_collision_details(self):
        return self._lib.orc_last_collision_details(self._h).decode()

    def batch_set_traj(self, bid, traj):
        t = _f64(traj)
        self._check(self._lib.orc_batch_set_traj(self._h, bid, _dp(t), t.size))

    def add_kinbody_boxes(self, name, boxes, transform=None):
        """boxes: list of (pose7, half_extents3) in the kinbody frame."""
        poses = _f64([b[0] for b in boxes]).reshape(-1, 7)
        halfs = _f64([b[1] for b in boxes]).reshape(-1, 3)
        self._check(self._lib.orc_env_add_kinbody_boxes(self._h, name.encode(), len(boxes), _dp(poses), _dp(halfs)))
        if transform is not None:
            self.set_kinbody_transform(name, transform)

    def add_kinbody_trimesh(self, name, triangles, transform=None):
        """triangles: [n][3][3] vertices in the kinbody frame (KinBody::InitFromTrimesh); added to a kinbody of that name if there is one"""
        tri = _f64(triangles).reshape(-1, 9)
        self._check(self._lib.orc_env_add_kinbody_trimesh(self._h, name.encode(), len(tri), _dp(tri)))
        if transform is not None:
            self.set_kinbody_transform(name, transform)

    def set_kinbody_transform(self, name, pose):
        self._check(self._lib.orc_kinbody_set_transform(self._h, name.encode(), _dp(_f64(pose))))

    def body_transform(self, name):
        """GetTransform of a robot or kinbody (a held kinbody: where its link carries it now)"""
        pose = np.zeros(7)
        self._check(self._lib.orc_body_get_transform(self._h, name.encode(), _dp(pose)))
        return pose

    def set_kinbody_spheres(self, name, pos, radius):
        """the <orcdchomp><spheres> data of a kinbody: what `create` reads from a body the robot holds"""
        pos = _f64(pos).reshape(-1, 3); radius = _f64(radius).reshape(-1)
        assert len(pos) == len(radius)
        self._check(self._lib.orc_kinbody_set_spheres(self._h, name.encode(), len(radius), _dp(pos), _dp(radius)))

    def grab(self, robot, kinbody, link):
        """RobotBase::Grab(body, link): link = robot link index"""
        self._check(self._lib.orc_robot_grab(self._h, robot.encode(), kinbody.encode(), int(link)))

    def release(self, robot, kinbody=None):
        """RobotBase::Release(body), or ReleaseAllGrabbed() without a body"""
        if kinbody is None:
            self._check(self._lib.orc_robot_release_all(self._h, robot.encode()))
        else:
            self._check(self._lib.orc_robot_release(self._h, robot.encode(), kinbody.encode()))

    def enable_kinbody(self, name, enabled=True):
        self._check(self._lib.orc_kinbody_enable(self._h, name.encode(), 1 if enabled else 0))

    # ---- fields ---------------------------------------------------------------------
    def add_sdf(self, kinbody, data, lengths, pose_kinbody_gsdf):
        data = _f64(data)
        sizes = np.asarray(data.shape, dtype=np.int32)
        self._check(self._lib.orc_scene_add_sdf(self._h, kinbody.encode(), _ip(sizes), _dp(_f64(lengths)),
                                                _dp(_f64(pose_kinbody_gsdf)), _dp(data)))

    def get_sdf(self, kinbody):
        sizes = np.zeros(3, dtype=np.int32); lengths = np.zeros(3); pose = np.zeros(7)
        self._check(self._lib.orc_scene_get_sdf(self._h, kinbody.encode(), _ip(sizes), _dp(lengths), _dp(pose), None, 0))
        data = np.zeros(tuple(int(s) for s in sizes))
        self._check(self._lib.orc_scene_get_sdf(self._h, kinbody.encode(), _ip(sizes), _dp(lengths), _dp(pose),
                                                _dp(data), data.size))
        return data, lengths, pose

    # ---- kernel-level batch API -------------------------------------------------------
    def batch_params(self, **kw):
        p = _capi.BatchParams()
        self._lib.orc_batch_params_default(C.byref(p))
        for k, v in kw.items():
            if k in ("lambda", "lambda_"):
                p.lambda_ = v
            elif k == "D":
                p.derivative = v
            else:
                if not hasattr(p, k):
                    raise TypeError("unknown batch parameter %s" % k)
                setattr(p, k, v)
        return p

    def batch_create(self, robot, goals, starts=None, basegoals=None, seeds=None, **params):
        p = self.batch_params(**params)
        goals = _f64(goals)
        goals = goals.reshape(1, -1) if goals.ndim == 1 else goals
        n_runs = goals.shape[0]
        st = None if starts is None else _f64(starts)
        bg = None if basegoals is None else _f64(basegoals)
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        bid = C.c_int(0)
        self._check(self._lib.orc_batch_create(
            self._h, robot.encode(), C.byref(p), n_runs,
            None if st is None else _dp(st), _dp(goals), None if bg is None else _dp(bg),
            None if sd is None else sd.ctypes.data_as(_capi.c_uint_p), C.byref(bid)))
        return bid.value

    def batch_dims(self, bid):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self._lib.orc_batch_dims(self._h, bid, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def batch_iterate(self, bid, n_iter):
        n_runs = self.batch_dims(bid)[0]
        costs = np.zeros((n_runs, 3)); status = np.zeros(n_runs, dtype=np.int32)
        self._check(self._lib.orc_batch_iterate(self._h, bid, n_iter, _dp(costs), _ip(status)))
        return costs, status

    def batch_iterate_async(self, bid, n_iter):
        self._check(self._lib.orc_batch_iterate_async(self._h, bid, n_iter))

    def batch_sync(self, bid, fetch=True):
        if not fetch:
            self._check(self._lib.orc_batch_sync(self._h, bid, None, None))
            return None
        n_runs = self.batch_dims(bid)[0]
        costs = np.zeros((n_runs, 3)); status = np.zeros(n_runs, dtype=np.int32)
        self._check(self._lib.orc_batch_sync(self._h, bid, _dp(costs), _ip(status)))
        return costs, status

    def batch_iterations_done(self, bid):
        """iterations every run completed in the last iterate call (n_iter unless it left its joint limits)"""
        n_runs = self.batch_dims(bid)[0]
        it = np.zeros(n_runs, dtype=np.int32)
        self._check(self._lib.orc_batch_iterations_done(self._h, bid, _ip(it)))
        return it

    def batch_trace(self, bid, n_iter):
        n_runs = self.batch_dims(bid)[0]
        tr = np.zeros((n_runs, n_iter, 3))
        self._check(self._lib.orc_batch_get_trace(self._h, bid, _dp(tr), tr.size))
        return tr

    def batch_set_noise(self, bid, noise):
        noise = _f64(noise)
        self._check(self._lib.orc_batch_set_noise(self._h, bid, _dp(noise), noise.shape[1]))

    def batch_gettraj(self, bid):
        n_runs, n_points, n = self.batch_dims(bid)
        out = np.zeros((n_runs, n_points, n))
        self._check(self._lib.orc_batch_gettraj(self._h, bid, _dp(out), out.size))
        return out

    def batch_collision_verdict(self, bid):
        """gettraj's collision re-check for every run of the batch, on the device: returns a dict of
        arrays per run: collides (0/1), time of the first contact on the retimed trajectory, XML index
        of the sphere, index of the field, penetration depth [m]"""
        n_runs = self.batch_dims(bid)[0]
        col = np.zeros(n_runs, dtype=np.int32); sph = np.zeros(n_runs, dtype=np.int32); fld = np.zeros(n_runs, dtype=np.int32)
        tim = np.zeros(n_runs); dep = np.zeros(n_runs)
        self._check(self._lib.orc_batch_collision_verdict(self._h, bid, _ip(col), _dp(tim), _ip(sph), _ip(fld), _dp(dep)))
        return dict(collides=col, time=tim, sphere=sph, field=fld, depth=dep)

    def batch_plan(self, bid):
        """what the planner chose for this batch (orc_batch_get_state "plan")"""
        out = np.zeros(8)
        self._check(self._lib.orc_batch_get_state(self._h, bid, b"plan", _dp(out), out.size))
        keys = ("variant", "threads", "lds_bytes", "tile_m", "solve_mode", "workgroups_per_cu", "tiles", "lanes_per_waypoint")
        return {k: int(v) for k, v in zip(keys, out)}

    def batch_state(self, bid, which):
        n_runs, n_points, n = self.batch_dims(bid)
        out = np.zeros((n_runs, n_points - 2, n))
        self._check(self._lib.orc_batch_get_state(self._h, bid, which.encode(), _dp(out), out.size))
        return out

    def batch_destroy(self, bid):
        self._check(self._lib.orc_batch_destroy(self._h, bid))

    def kernel_time(self, reset=False):
        ms = C.c_double(); n = C.c_int()
        self._check(self._lib.orc_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value
