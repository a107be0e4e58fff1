"""ctypes binding of include/orcdchomp_amd.h (liborcdchomp_amd.so, built in-tree).

The library is the product: there is no Python or CPU fallback.  Importing this
module only loads the shared object; a GPU is needed as soon as a Module is made.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORC_LIB") or os.path.join(_HERE, "liborcdchomp_amd.so")   # ORC_LIB: diagnostic builds

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_uint_p = C.POINTER(C.c_uint)


class RobotDesc(C.Structure):
    _fields_ = [("n_links", C.c_int), ("parent", c_int_p), ("pose_parent_joint", c_double_p),
                ("joint_type", c_int_p), ("axis", c_double_p), ("dof_index", c_int_p),
                ("n_dof", C.c_int), ("limit_lower", c_double_p), ("limit_upper", c_double_p),
                ("n_spheres", C.c_int), ("sphere_link", c_int_p), ("sphere_pos", c_double_p),
                ("sphere_radius", c_double_p)]


class BatchParams(C.Structure):
    _fields_ = [("n_points", C.c_int), ("floating_base", C.c_int), ("lambda_", C.c_double),
                ("derivative", C.c_int), ("use_momentum", C.c_int), ("use_hmc", C.c_int),
                ("hmc_resample_lambda", C.c_double), ("epsilon", C.c_double),
                ("epsilon_self", C.c_double), ("obs_factor", C.c_double),
                ("obs_factor_self", C.c_double), ("precision", C.c_int)]


# every symbol include/orcdchomp_amd.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("orc_module_new", C.c_void_p, [C.c_int]),
    ("orc_module_new_multi", C.c_void_p, [c_int_p, C.c_int]),
    ("orc_module_free", None, [C.c_void_p]),
    ("orc_last_error", C.c_char_p, [C.c_void_p]),
    ("orc_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("orc_set_num_streams", C.c_int, [C.c_void_p, C.c_int]),
    ("orc_send_command", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("orc_last_reply_size", C.c_size_t, [C.c_void_p]),
    ("orc_last_reply", C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    ("orc_env_add_robot", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(RobotDesc)]),
    ("orc_robot_set_transform", C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    ("orc_robot_set_dof_values", C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int]),
    ("orc_robot_set_active_dofs", C.c_int, [C.c_void_p, C.c_char_p, c_int_p, C.c_int]),
    ("orc_robot_set_velocity_limits", C.c_int, [C.c_void_p, C.c_char_p, c_double_p, C.c_int]),
    ("orc_set_workgroup_threads", C.c_int, [C.c_void_p, C.c_int]),
    ("orc_set_workgroups_per_cu", C.c_int, [C.c_void_p, C.c_int]),
    ("orc_robot_set_link_names", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int]),
    ("orc_robot_add_manipulator", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, c_double_p]),
    ("orc_robot_set_adjacent_links", C.c_int, [C.c_void_p, C.c_char_p, c_int_p, C.c_int]),
    ("orc_robot_set_active_manipulator", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    ("orc_robot_set_self_check", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("orc_env_add_kinbody_boxes", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_double_p, c_double_p]),
    ("orc_env_add_kinbody_trimesh", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_double_p]),
    ("orc_kinbody_set_transform", C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    ("orc_kinbody_enable", C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    ("orc_body_get_transform", C.c_int, [C.c_void_p, C.c_char_p, c_double_p]),
    ("orc_kinbody_set_spheres", C.c_int, [C.c_void_p, C.c_char_p, C.c_int, c_double_p, c_double_p]),
    ("orc_robot_grab", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int]),
    ("orc_robot_release", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    ("orc_robot_release_all", C.c_int, [C.c_void_p, C.c_char_p]),
    ("orc_scene_add_sdf", C.c_int, [C.c_void_p, C.c_char_p, c_int_p, c_double_p, c_double_p, c_double_p]),
    ("orc_scene_get_sdf", C.c_int, [C.c_void_p, C.c_char_p, c_int_p, c_double_p, c_double_p, c_double_p, C.c_size_t]),
    ("orc_batch_params_default", None, [C.POINTER(BatchParams)]),
    ("orc_batch_create", C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(BatchParams), C.c_int, c_double_p,
                                   c_double_p, c_double_p, c_uint_p, c_int_p]),
    ("orc_batch_iterate", C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p, c_int_p]),
    ("orc_batch_iterate_async", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("orc_batch_sync", C.c_int, [C.c_void_p, C.c_int, c_double_p, c_int_p]),
    ("orc_batch_iterations_done", C.c_int, [C.c_void_p, C.c_int, c_int_p]),
    ("orc_batch_get_trace", C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_size_t]),
    ("orc_batch_set_noise", C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_int]),
    ("orc_batch_gettraj", C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_size_t]),
    ("orc_batch_collision_verdict", C.c_int, [C.c_void_p, C.c_int, c_int_p, c_double_p, c_int_p, c_int_p, c_double_p]),
    ("orc_batch_get_state", C.c_int, [C.c_void_p, C.c_int, C.c_char_p, c_double_p, C.c_size_t]),
    ("orc_batch_dims", C.c_int, [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    ("orc_batch_set_traj", C.c_int, [C.c_void_p, C.c_int, c_double_p, C.c_size_t]),
    ("orc_last_collision_details", C.c_char_p, [C.c_void_p]),
    ("orc_batch_destroy", C.c_int, [C.c_void_p, C.c_int]),
    ("orc_kernel_time", C.c_int, [C.c_void_p, c_double_p, c_int_p, C.c_int]),
    ("orc_host_bin_sdf", C.c_int, [c_int_p, c_double_p, c_double_p, c_double_p]),
    ("orc_host_flood_fill", C.c_int, [c_int_p, c_double_p, C.c_size_t]),
    ("orc_host_voxelize_boxes", C.c_int, [c_int_p, c_double_p, c_double_p, C.c_double, C.c_int, c_double_p, c_double_p,
                                          c_double_p]),
    ("orc_host_voxelize_trimesh", C.c_int, [c_int_p, c_double_p, c_double_p, C.c_double, C.c_int, c_double_p, c_double_p]),
    ("orc_host_shparse", C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    ("orc_host_metric", C.c_int, [C.c_int, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                                  c_double_p, C.c_int, c_double_p]),
    ("orc_host_metric_free_start", C.c_int, [C.c_int, C.c_int, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p,
                                  c_double_p, C.c_int, c_double_p]),
    ("orc_host_metric_semisep_rank", C.c_int, [C.c_int, C.c_int, C.c_double, C.c_int]),
    ("orc_host_gsl_stream", C.c_int, [C.c_ulong, C.c_double, C.c_int, c_double_p, c_double_p]),
]

_LIB = None


def csrc_hash():
    """fingerprint of the kernel and host sources the library is built from (csrc/*, the header): profiles carry it so that
    counters taken from another build are not quoted for this one (bench.py, scripts/summarize_profile.py)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(_HERE, "csrc", "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build():
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liborcdchomp_amd.so is missing: run __graft_entry__.build() "
                "(or `make -C or_cdchomp_amd/csrc`); there is no fallback path")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
