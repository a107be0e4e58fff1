"""ctypes front end for oracle/liboracle.so and oracle/_ref/libcd_ref.so.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


def build(ref=True):
    """Compile liboracle.so (always) and _ref/libcd_ref.so (when the reference tree is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isdir("/root/reference/src/libcd"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class Grid(C.Structure):
    _fields_ = [("n", C.c_int), ("sizes", C.c_int * 3), ("ncells", C.c_size_t),
                ("lengths", C.c_double * 3), ("data", c_double_p)]


class Grabbed(C.Structure):
    _fields_ = [("robot_link", C.c_int), ("pose_world_klink", C.c_double * 7), ("n_spheres", C.c_int),
                ("sphere_pos", c_double_p), ("sphere_radius", c_double_p),
                ("has_grab_state", C.c_int), ("grab_base_pose", C.c_double * 7), ("grab_dofvals", c_double_p)]


class Robot(C.Structure):
    _fields_ = [("n_links", C.c_int), ("parent", c_int_p), ("pose_parent_joint", c_double_p),
                ("joint_type", c_int_p), ("axis", c_double_p), ("dof_index", c_int_p),
                ("n_dof", C.c_int), ("limit_lower", c_double_p), ("limit_upper", c_double_p),
                ("n_spheres", C.c_int), ("sphere_link", c_int_p), ("sphere_pos", c_double_p),
                ("sphere_radius", c_double_p), ("n_adjacent", C.c_int), ("adjacent", c_int_p),
                ("n_grabbed", C.c_int), ("grabbed", C.POINTER(Grabbed))]


class RunParams(C.Structure):
    _fields_ = [("n_points", C.c_int), ("floating_base", C.c_int), ("lambda_", C.c_double),
                ("D", C.c_int), ("use_momentum", C.c_int), ("use_hmc", C.c_int),
                ("hmc_resample_lambda", C.c_double), ("seed", C.c_uint),
                ("epsilon", C.c_double), ("epsilon_self", C.c_double),
                ("obs_factor", C.c_double), ("obs_factor_self", C.c_double),
                ("start_tsr", C.c_int), ("start_ee_link", C.c_int), ("start_tool", C.c_double * 7),
                ("start_T0w", C.c_double * 7), ("start_Twe", C.c_double * 7), ("start_Bw", C.c_double * 12)]


class Rng(C.Structure):
    _fields_ = [("mt", C.c_ulong * 624), ("mti", C.c_int)]


class Chomp(C.Structure):
    _fields_ = [("m", C.c_int), ("n", C.c_int), ("T", c_double_p), ("ldt", C.c_int),
                ("T_points", C.c_void_p), ("G", c_double_p), ("G_points", C.c_void_p),
                ("AG", c_double_p), ("AG_points", C.c_void_p), ("D", C.c_int),
                ("wds", c_double_p), ("initsfinals", c_double_p), ("inits", C.c_void_p),
                ("finals", C.c_void_p), ("dt", C.c_double), ("A", c_double_p), ("Ainv", c_double_p),
                ("B", c_double_p), ("trC", C.c_double), ("jlimit_lower", c_double_p),
                ("jlimit_upper", c_double_p), ("Gjlimit", c_double_p), ("GjlimitAinv", c_double_p),
                ("cost_nxn", c_double_p), ("cost_mxn", c_double_p), ("Kvels", c_double_p),
                ("Evels", c_double_p), ("vels", c_double_p), ("cptr", C.c_void_p),
                ("cost_pre", C.c_void_p), ("cost", C.c_void_p), ("lambda_", C.c_double),
                ("use_momentum", C.c_int), ("leapfrog_first", C.c_int), ("last_num_limadjs", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.ora_grid_create.restype = C.POINTER(Grid)
        L.ora_grid_create.argtypes = [c_int_p, c_double_p, C.c_double]
        L.ora_grid_free.argtypes = [C.POINTER(Grid)]
        L.ora_grid_lookup_index.argtypes = [C.POINTER(Grid), c_double_p, C.POINTER(C.c_size_t)]
        L.ora_grid_double_interp.argtypes = [C.POINTER(Grid), c_double_p, c_double_p]
        L.ora_grid_double_grad.argtypes = [C.POINTER(Grid), c_double_p, c_double_p]
        L.ora_grid_double_bin_sdf.argtypes = [C.POINTER(C.POINTER(Grid)), C.POINTER(Grid)]
        L.ora_grid_double_dt_sqeuc.argtypes = [C.POINTER(C.POINTER(Grid)), C.POINTER(Grid)]
        L.ora_grid_flood_fill_1_to_0.restype = C.c_long
        L.ora_grid_flood_fill_1_to_0.argtypes = [C.POINTER(Grid), C.c_size_t]
        L.ora_grid_center_index.argtypes = [C.POINTER(Grid), C.c_size_t, c_double_p]
        for nm in ("ora_kin_pose_compose", "ora_kin_pose_compos", "ora_kin_pose_compose_vec"):
            getattr(L, nm).argtypes = [c_double_p, c_double_p, c_double_p]
        L.ora_kin_pose_invert.argtypes = [c_double_p, c_double_p]
        L.ora_kin_pose_normalize.argtypes = [c_double_p]
        L.ora_spatial_pose_jac.argtypes = [c_double_p, c_double_p]
        L.ora_spatial_xm_from_pose.argtypes = [c_double_p, c_double_p]
        L.ora_rng_set.argtypes = [C.POINTER(Rng), C.c_ulong]
        L.ora_rng_get.restype = C.c_ulong
        L.ora_rng_get.argtypes = [C.POINTER(Rng)]
        L.ora_rng_uniform.restype = C.c_double
        L.ora_rng_uniform.argtypes = [C.POINTER(Rng)]
        L.ora_ran_gaussian.restype = C.c_double
        L.ora_ran_gaussian.argtypes = [C.POINTER(Rng), C.c_double]
        L.ora_chomp_create.argtypes = [C.POINTER(C.POINTER(Chomp)), C.c_int, C.c_int, C.c_int, c_double_p, C.c_int]
        L.ora_chomp_init.argtypes = [C.POINTER(Chomp)]
        L.ora_chomp_free.argtypes = [C.POINTER(Chomp)]
        L.ora_chomp_iterate.argtypes = [C.POINTER(Chomp), C.c_int, c_double_p, c_double_p, c_double_p]
        L.ora_robot_fk.argtypes = [C.POINTER(Robot), c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]
        L.ora_run_params_default.argtypes = [C.POINTER(RunParams)]
        L.ora_run_create.restype = C.c_void_p
        L.ora_run_create.argtypes = [C.POINTER(Robot), c_double_p, c_double_p, C.c_int, c_int_p, c_double_p,
                                     c_double_p, C.c_int, C.POINTER(C.POINTER(Grid)), c_double_p,
                                     C.POINTER(RunParams), C.POINTER(C.c_char_p)]
        L.ora_run_iterate.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p]
        L.ora_run_iterate_noise.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int]
        L.ora_run_destroy.argtypes = [C.c_void_p]
        L.ora_run_set_traj.argtypes = [C.c_void_p, c_double_p]
        L.ora_run_collision_recheck.argtypes = [C.c_void_p, c_double_p, c_int_p, c_double_p, c_int_p, c_int_p, c_double_p]
        L.ora_sample_starttraj.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int, c_double_p]
        L.ora_sample_starttraj_floating.argtypes = [C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p]
        L.ora_gettraj_affine_groups.argtypes = [c_double_p, C.c_int, C.c_int, c_double_p, c_double_p]
        L.ora_robot_self_pairs.argtypes = [C.POINTER(Robot), C.POINTER(C.c_ubyte)]
        for nm in ("ora_run_n", "ora_run_m", "ora_run_n_points", "ora_run_n_spheres_active",
                   "ora_run_n_spheres", "ora_run_hmc_resample_iter", "ora_run_iter"):
            getattr(L, nm).argtypes = [C.c_void_p]
        L.ora_run_traj.restype = c_double_p
        L.ora_run_traj.argtypes = [C.c_void_p]
        L.ora_run_chomp.restype = C.POINTER(Chomp)
        L.ora_run_chomp.argtypes = [C.c_void_p]
        L.ora_run_eval_obstacle.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
        L.ora_run_add_contsr.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        L.ora_run_eval_contsr.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]
        L.ora_kin_pose_from_dR.argtypes = [c_double_p, c_double_p, c_double_p]
        L.ora_kin_pose_to_xyzypr.argtypes = [c_double_p, c_double_p]
        L.ora_run_sphere_order.argtypes = [C.c_void_p, c_int_p]
        L.ora_run_self_excluded.argtypes = [C.c_void_p, C.POINTER(C.c_ubyte)]
        L.ora_batch_run.argtypes = [C.POINTER(Robot), c_double_p, c_double_p, C.c_int, c_int_p, C.c_int,
                                    c_double_p, c_double_p, C.c_int, C.POINTER(C.POINTER(Grid)), c_double_p,
                                    C.POINTER(RunParams), C.POINTER(C.c_uint), C.c_int,
                                    c_double_p, c_double_p, c_int_p, C.c_int]
        L.ora_util_shparse.argtypes = [C.c_char_p, c_int_p, C.POINTER(C.POINTER(C.c_char_p))]
        _LIB = L
    return _LIB


def ref():
    """The reference's own grid/flood/shparse code (oracle/_ref), or None when absent."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libcd_ref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def dp(a):
    return a.ctypes.data_as(c_double_p)


def ip(a):
    return a.ctypes.data_as(c_int_p)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


# ------------------------------------------------------------------ grid helpers
class OraGrid:
    """Owns an ora_grid whose cells are a numpy array (C order [x][y][z])."""

    def __init__(self, data, lengths):
        self.data = np.ascontiguousarray(data, dtype=np.float64)
        assert self.data.ndim == 3
        self.g = Grid()
        self.g.n = 3
        for i in range(3):
            self.g.sizes[i] = self.data.shape[i]
            self.g.lengths[i] = float(lengths[i])
        self.g.ncells = self.data.size
        self.g.data = dp(self.data)

    @property
    def ptr(self):
        return C.pointer(self.g)

    def interp(self, p):
        p = f64(p)
        v = C.c_double()
        err = lib().ora_grid_double_interp(self.ptr, dp(p), C.byref(v))
        return err, v.value

    def grad(self, p):
        p = f64(p)
        g = np.zeros(3)
        err = lib().ora_grid_double_grad(self.ptr, dp(p), dp(g))
        return err, g

    def bin_sdf(self):
        out = C.POINTER(Grid)()
        err = lib().ora_grid_double_bin_sdf(C.byref(out), self.ptr)
        assert err == 0
        arr = np.ctypeslib.as_array(out.contents.data, shape=self.data.shape).copy()
        lib().ora_grid_free(out)
        return OraGrid(arr, [self.g.lengths[i] for i in range(3)])

    def flood_fill(self, index_start=0):
        return lib().ora_grid_flood_fill_1_to_0(self.ptr, index_start)


# ------------------------------------------------------------------ robot helpers
class OraRobot:
    def __init__(self, model, grabbed=()):
        """grabbed: the kinbodies the robot holds, in GetGrabbed() order (reference src/orcdchomp_mod.cpp:2168-2171):
        tuples (robot link index, pose_world_klink[7], sphere_pos[k][3], sphere_radius[k]) and, optionally, two more entries
        (base_pose[7], dofvals[n_dof]): the robot's state at the moment of the grab (default: the state of create)"""
        a = model.arrays()
        self._keep = a
        self._grab_keep = []
        self._grabbed = (Grabbed * max(len(grabbed), 1))()
        for gi, entry in enumerate(grabbed):
            link, pose, pos, rad = entry[:4]
            pos = f64(pos).reshape(-1, 3); rad = f64(rad).reshape(-1)
            assert len(pos) == len(rad)
            self._grab_keep.append((pos, rad))
            g = self._grabbed[gi]
            g.has_grab_state = 1 if len(entry) > 4 else 0
            if len(entry) > 4:
                gd = f64(entry[5]).reshape(-1)
                assert len(gd) == a["n_dof"]
                self._grab_keep.append(gd)
                for i in range(7):
                    g.grab_base_pose[i] = float(entry[4][i])
                g.grab_dofvals = dp(gd)
            g.robot_link = int(link)
            for i in range(7):
                g.pose_world_klink[i] = float(pose[i])
            g.n_spheres = len(rad)
            g.sphere_pos = dp(pos); g.sphere_radius = dp(rad)
        r = Robot()
        r.n_links = a["n_links"]
        r.parent = ip(a["parent"])
        r.pose_parent_joint = dp(a["pose_parent_joint"])
        r.joint_type = ip(a["joint_type"])
        r.axis = dp(a["axis"])
        r.dof_index = ip(a["dof_index"])
        r.n_dof = a["n_dof"]
        r.limit_lower = dp(a["limit_lower"])
        r.limit_upper = dp(a["limit_upper"])
        r.n_spheres = a["n_spheres"]
        r.sphere_link = ip(a["sphere_link"])
        r.sphere_pos = dp(a["sphere_pos"])
        r.sphere_radius = dp(a["sphere_radius"])
        r.n_adjacent = a["n_adjacent"]
        r.adjacent = ip(a["adjacent"])
        r.n_grabbed = len(grabbed)
        r.grabbed = C.cast(self._grabbed, C.POINTER(Grabbed))
        self.r = r
        self.model = model

    @property
    def ptr(self):
        return C.pointer(self.r)

    def fk(self, base_pose, dofvals):
        n = self.r.n_links
        R = np.zeros((n, 9)); t = np.zeros((n, 3)); ax = np.zeros((n, 3)); an = np.zeros((n, 3))
        lib().ora_robot_fk(self.ptr, dp(f64(base_pose)), dp(f64(dofvals)), dp(R), dp(t), dp(ax), dp(an))
        return R.reshape(n, 3, 3), t, ax, an


def default_params(start_tsr=None, **kw):
    """start_tsr = (ee_link, tool[7], T0w[7], Twe[7], Bw[6][2]): `start_tsr` of create"""
    p = RunParams()
    lib().ora_run_params_default(C.byref(p))
    known = set(f[0] for f in RunParams._fields_)
    for k, v in kw.items():
        if k == "lambda":
            k = "lambda_"
        if k not in known:
            raise TypeError("oracle run parameter %r does not exist" % k)
        setattr(p, k, v)
    if start_tsr is not None:
        ee_link, tool, T0w, Twe, Bw = start_tsr
        p.start_tsr = 1
        p.start_ee_link = int(ee_link)
        for i in range(7):
            p.start_tool[i] = float(tool[i]); p.start_T0w[i] = float(T0w[i]); p.start_Twe[i] = float(Twe[i])
        for i, v in enumerate(np.asarray(Bw, dtype=float).reshape(12)):
            p.start_Bw[i] = float(v)
    return p


class OraRun:
    def __init__(self, robot, base_pose, dofvals, adofindices, adofgoal, grids, poses_world_gsdf,
                 params, basegoal=None):
        self.robot = robot
        self.grids = list(grids)
        self._gp = (C.POINTER(Grid) * len(self.grids))(*[g.ptr for g in self.grids])
        self._poses = f64(poses_world_gsdf).reshape(-1, 7)
        self._adof = np.ascontiguousarray(adofindices, dtype=np.int32)
        err = C.c_char_p()
        bg = None if basegoal is None else dp(f64(basegoal))
        self.h = lib().ora_run_create(robot.ptr, dp(f64(base_pose)), dp(f64(dofvals)), len(self._adof),
                                      ip(self._adof), dp(f64(adofgoal)), bg, len(self.grids), self._gp,
                                      dp(self._poses), C.byref(params), C.byref(err))
        if not self.h:
            raise RuntimeError(err.value.decode())
        L = lib()
        self.n = L.ora_run_n(self.h)
        self.m = L.ora_run_m(self.h)
        self.n_points = L.ora_run_n_points(self.h)
        self.Sa = L.ora_run_n_spheres_active(self.h)
        self.S = L.ora_run_n_spheres(self.h)

    def traj(self):
        return np.ctypeslib.as_array(lib().ora_run_traj(self.h), shape=(self.n_points, self.n))

    def chomp(self):
        return lib().ora_run_chomp(self.h).contents

    def mat(self, name, rows, cols):
        return np.ctypeslib.as_array(getattr(self.chomp(), name), shape=(rows, cols))

    def iterate(self, n_iter, trace=False, noise=None):
        costs = np.zeros(3)
        tr = np.zeros((max(n_iter, 1), 3)) if trace else None
        if noise is None:
            st = lib().ora_run_iterate(self.h, n_iter, dp(costs), dp(tr) if trace else None)
        else:
            noise = f64(noise)
            st = lib().ora_run_iterate_noise(self.h, n_iter, dp(costs), dp(tr) if trace else None,
                                             dp(noise), noise.shape[0])
        return (st, costs, tr) if trace else (st, costs)

    def iter(self):
        return lib().ora_run_iter(self.h)

    def set_traj(self, traj):
        t = f64(traj)
        assert t.shape == (self.n_points, self.n)
        lib().ora_run_set_traj(self.h, dp(t))

    def collision_recheck(self, vmax):
        """gettraj's re-check (reference src/orcdchomp_mod.cpp:2958-3006): dict(collides, time, sphere, field, depth, samples)"""
        col = C.c_int(); sph = C.c_int(); fld = C.c_int(); tim = C.c_double(); dep = C.c_double()
        n = lib().ora_run_collision_recheck(self.h, dp(f64(vmax)), C.byref(col), C.byref(tim), C.byref(sph), C.byref(fld),
                                            C.byref(dep))
        return dict(collides=col.value, time=tim.value, sphere=sph.value, field=fld.value, depth=dep.value, samples=n)

    def add_contsr(self, ee_link, tool, T0w, Twe, Bw):
        """`con_tsr all ...` / `everyn_tsr` (reference src/orcdchomp_mod.cpp:2466-2480, 2582-2612): a TSR hard
        constraint on every moving point; Bw [6][2].  Returns the constraint's dimension k."""
        err = lib().ora_run_add_contsr(self.h, int(ee_link), dp(f64(tool)), dp(f64(T0w)), dp(f64(Twe)), dp(f64(Bw).reshape(12)))
        assert err == 0
        return int(sum(1 for row in np.asarray(Bw, dtype=float).reshape(6, 2) if row[0] == 0.0 and row[1] == 0.0))

    def eval_contsr(self, which, point):
        h = np.zeros(6); J = np.zeros((6, self.n))
        k = lib().ora_run_eval_contsr(self.h, which, dp(f64(point)), dp(h), dp(J))
        return h[:k].copy(), J[:k].copy()

    def eval_obstacle(self):
        G = np.zeros((self.m, self.n)); costs = np.zeros(self.m)
        P = np.zeros((self.n_points, self.Sa, 3))
        lib().ora_run_eval_obstacle(self.h, dp(G), dp(costs), dp(P))
        return G, costs, P

    def sphere_order(self):
        idx = np.zeros(self.S, dtype=np.int32)
        lib().ora_run_sphere_order(self.h, ip(idx))
        return idx

    def self_excluded(self):
        """[S][S]: pairs of the run's spheres the re-check's self-collision leg never tests"""
        out = np.zeros((self.S, self.S), dtype=np.uint8)
        lib().ora_run_self_excluded(self.h, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out

    def destroy(self):
        if self.h:
            lib().ora_run_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def pose_from_dR(d, R):
    """cd_kin_pose_from_dR (reference src/libcd/kin.c:510-517): [x y z qx qy qz qw] from a translation and a 3x3 rotation"""
    pose = np.zeros(7)
    lib().ora_kin_pose_from_dR(dp(pose), dp(f64(d)), dp(f64(R).reshape(9)))
    return pose


def sample_starttraj(wp, deltatime, n_points):
    """create's starttraj sampling (reference src/orcdchomp_mod.cpp:2375-2416)"""
    wp = f64(wp); dtm = f64(deltatime)
    out = np.zeros((n_points, wp.shape[1]))
    lib().ora_sample_starttraj(wp.shape[0], wp.shape[1], dp(wp), dp(dtm), n_points, dp(out))
    return out


def sample_starttraj_floating(wp_joint, wp_base, deltatime, n_points):
    """create's starttraj sampling with floating_base (reference src/orcdchomp_mod.cpp:2378-2404): wp_base rows in
    OpenRAVE's order x y z qw qx qy qz; returns [n_points][7 + n_adof] with the base in libcd's order, normalised"""
    wj = f64(wp_joint); wb = f64(wp_base); dtm = f64(deltatime)
    out = np.zeros((n_points, 7 + wj.shape[1]))
    lib().ora_sample_starttraj_floating(wj.shape[0], wj.shape[1], dp(wj), dp(wb), dp(dtm), n_points, dp(out))
    return out


def gettraj_affine_groups(traj, deltatime):
    """gettraj's affine_transform / affine_velocities rows of a floating-base run (reference
    src/orcdchomp_mod.cpp:2912-2949): [n_points][15] = deltatime, x y z qw qx qy qz, their velocities"""
    t = f64(traj); dtm = f64(deltatime)
    out = np.zeros((t.shape[0], 15))
    lib().ora_gettraj_affine_groups(dp(t), t.shape[0], t.shape[1], dp(dtm), dp(out))
    return out


def self_pairs_excluded(robot):
    """[n_links][n_links] 0/1: link pairs the self-collision leg of gettraj's re-check skips"""
    nl = robot.r.n_links
    out = np.zeros((nl, nl), dtype=np.uint8)
    lib().ora_robot_self_pairs(robot.ptr, out.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return out


def batch_run(robot, base_pose, dofvals, adofindices, adofgoals, grids, poses_world_gsdf, params,
              n_iter, basegoals=None, seeds=None, max_threads=0):
    adof = np.ascontiguousarray(adofindices, dtype=np.int32)
    goals = f64(adofgoals).reshape(-1, len(adof))
    n_runs = goals.shape[0]
    n = (7 if params.floating_base else 0) + len(adof)
    gp = (C.POINTER(Grid) * len(grids))(*[g.ptr for g in grids])
    poses = f64(poses_world_gsdf).reshape(-1, 7)
    traj = np.zeros((n_runs, params.n_points, n)); costs = np.zeros((n_runs, 3))
    status = np.zeros(n_runs, dtype=np.int32)
    bg = None if basegoals is None else dp(f64(basegoals))
    sd = None
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        sd = seeds.ctypes.data_as(C.POINTER(C.c_uint))
    threads = lib().ora_batch_run(robot.ptr, dp(f64(base_pose)), dp(f64(dofvals)), len(adof), ip(adof), n_runs,
                                  dp(goals), bg, len(grids), gp, dp(poses), C.byref(params), sd, n_iter,
                                  dp(traj), dp(costs), ip(status), max_threads)
    return traj, costs, status, threads


def reference_sphere_list(bodies):
    """TEST INFRASTRUCTURE: the order of a run's sphere list as mod::create builds it (src/orcdchomp_mod.cpp:2168-2300), restated
    literally with Python lists.  bodies: the robot first, then the grabbed kinbodies in GetGrabbed() order; each a dict with
      xml     the <orcdchomp> spheres' ids in the order of the XML file,
      group   the ids of the "spheres" geometry-group spheres in (link index, geometry index) order (OpenRAVE >= 0.9, :2214-2259),
      active  the set of ids on links an active dof moves (:2266-2271).
    Returns (ids of the list from its head: active ones, then the inactive ones appended, :2299)."""
    run_list = []            # r->spheres, head first
    inactive = []            # s_inactive_head, head first
    for body in bodies:
        # the kdata list was built by head insertion while the XML was parsed (src/orcdchomp_kdata.cpp:90-94)
        kdata = []
        for sid in body["xml"]:
            kdata.insert(0, sid)
        k_spheres = []
        for sid in kdata:                        # :2178-2211  for (sel=d->sphereelems; sel; sel=sel->next) ... push_back
            k_spheres.append(sid)
        for sid in body.get("group", []):       # :2216-2259  links in index order, geometries in index order ... push_back
            k_spheres.append(sid)
        if not k_spheres:
            raise RuntimeError("no spheres! kinbody does not have a <orcdchomp> tag defined?")      # :2262-2263
        for sid in k_spheres:                    # :2265-2291
            if sid in body["active"]:
                run_list.insert(0, sid)          # "active; insert at head of r->spheres"
            else:
                inactive.insert(0, sid)          # "inactive; insert into s_inactive_head"
    return run_list + inactive                   # :2299
