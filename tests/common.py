"""Shared problem builders for the tests: the same inputs go to the oracle and to the product."""
import math

import numpy as np

from or_cdchomp_amd import robots, scenes

HUGE = np.inf


def grid_dims(aabb_pos, aabb_ext, cube_extent=0.02, aabb_padding=0.2):
    """sizes / lengths / grid pose of computedistancefield (reference src/orcdchomp_mod.cpp:386-409)."""
    sizes = [int(math.ceil((aabb_ext[i] + aabb_padding) / cube_extent)) for i in range(3)]
    lengths = [sizes[i] * 2.0 * cube_extent for i in range(3)]
    pose = [aabb_pos[i] - 0.5 * lengths[i] for i in range(3)] + [0.0, 0.0, 0.0, 1.0]
    return sizes, lengths, pose


def voxelize_axis_aligned(sizes, lengths, origin, boxes, cube_extent):
    """occupancy 1.0 free / inf hit for axis-aligned boxes given as (center, half) in the world;
    a cube of half-extent cube_extent at every cell centre, overlap deeper than 1e-9 counts."""
    occ = np.ones(sizes)
    axes = []
    for d in range(3):
        sub = np.arange(sizes[d])
        axes.append((0.5 + sub) / sizes[d] * lengths[d] + origin[d])
    for center, half in boxes:
        m = [np.abs(axes[d] - center[d]) < (cube_extent + half[d] - 1e-9) for d in range(3)]
        occ[np.ix_(m[0], m[1], m[2])] = HUGE
    return occ


def tabletop_problem(oracle_py, cube_extent=0.02, aabb_padding=0.2):
    """The synthetic tabletop of SURVEY.md 8d config 1: occupancy -> flood fill -> sdf, all by the oracle."""
    boxes = scenes.tabletop_boxes()
    tpose, thalf = boxes["table"][0]
    # AABB the way KinBodyComputeEnabledAABB forms it (reference src/orcdchomp_mod.cpp:103-137):
    # min/max corners first, then pos = (min+max)/2 and extents = max - pos
    lo = [tpose[i] - thalf[i] for i in range(3)]
    hi = [tpose[i] + thalf[i] for i in range(3)]
    apos = [0.5 * (lo[i] + hi[i]) for i in range(3)]
    aext = [hi[i] - apos[i] for i in range(3)]
    sizes, lengths, pose = grid_dims(apos, aext, cube_extent, aabb_padding)
    world = [(b[0][:3], b[1]) for name in boxes for b in boxes[name]]
    occ = voxelize_axis_aligned(sizes, lengths, pose[:3], world, cube_extent)
    g = oracle_py.OraGrid(occ, lengths)
    g.flood_fill(0)
    g.data[g.data == 1.0] = HUGE
    sdf = g.bin_sdf()
    return dict(sizes=sizes, lengths=lengths, pose=pose, occ=occ, sdf=sdf)


def wam_state():
    model = robots.wam7()
    dofvals = np.zeros(model.n_dof)
    dofvals[:7] = robots.WAM_START
    return model, list(robots.WAM_BASE_POSE), dofvals, list(range(7))


def wam_goals(n_runs, seed=20250101):
    """q_goal ~ U(lower+0.1, upper-0.1) per joint, run-major."""
    model = robots.wam7()
    lo = np.asarray(model.limit_lower[:7]) + 0.1
    hi = np.asarray(model.limit_upper[:7]) - 0.1
    rng = np.random.default_rng(seed)
    return rng.uniform(lo, hi, size=(n_runs, 7))


def setup_product_wam(mod, name="BarrettWAM"):
    """robot + tabletop bodies + computedistancefield on the product module."""
    model, base, dofvals, adofs = wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    scenes.add_tabletop(mod)
    mod.SendCommand("computedistancefield kinbody table")
    return model


# A run whose oracle result moves by `amp` (relative L2) when its goal moves by one ulp is "ill-conditioned": CHOMP
# bouncing off a joint limit amplifies a rounding-level difference ~4x per projection, and a rounding difference of
# the HIP path is amplified like any other.  Such a run is held to CHAOS_FACTOR x amp instead of the 1e-6 bar, with amp
# the larger of the two movements under goal x (1 +- 2^-52).  Measured (scripts/chaos_ratio.py, profiles/r04_chaos_ratio.txt:
# 2048 runs of config 2 and 512 of config 4, 180 of them ill-conditioned): err / amp has median 0.95, p90 3.9, p99 31,
# maximum 132 -- the ratio of two random projections on the run's unstable direction, heavy-tailed; with one
# perturbation instead of two the maximum is 240.  300 leaves a factor of two over the largest ratio seen (5000 until round 3).
CHAOS_FACTOR = 300.0
ULPS = (1.0 + 2.0 ** -52, 1.0 - 2.0 ** -52)


def amplification(run, goals, base):
    """how far each oracle run moves under one-ulp changes of its goal: `run(goals)` -> (traj, costs, status, ...), `base` its
    result on the goals themselves.  Returns (amp [n_runs], stable [n_runs]: the status is the same in all three)"""
    amp = np.zeros(len(goals))
    stable = np.ones(len(goals), dtype=bool)
    for f in ULPS:
        res = run(np.asarray(goals) * f)
        amp = np.maximum(amp, [rel_l2(res[0][k], base[0][k]) for k in range(len(goals))])
        stable &= (np.asarray(res[2]) == np.asarray(base[2]))
    return amp, stable


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---------------------------------------------------------------------------------------------
# BASELINE.json configurations at their full size, shared by the -m gpu tests,
# bench.py and tests/golden/make_fullsize_golden.py
CONFIG2_KW = dict(n_points=100, lambda_=100.0, obs_factor=500.0)


def config3_goals(rank=0, world=8, n_total=65536, seed=20250102):
    """config 3: 65 536 WAM goals drawn run-major with one seed, cut into contiguous blocks per GPU"""
    from or_cdchomp_amd import sharding
    lo, hi = sharding.shard_bounds(n_total, rank, world)
    return wam_goals(n_total, seed=seed)[lo:hi]


def config4_problem(n_runs=4096):
    """config 4: floating base + WAM arm, n=14, n_points=200, momentum + hmc, seed = run index"""
    _, base, _, _ = wam_state()
    rng = np.random.default_rng(20250103)
    goals = wam_goals(n_runs, seed=20250103)
    basegoals = np.tile(np.asarray(base, dtype=np.float64), (n_runs, 1))
    basegoals[:, :3] += rng.uniform(-0.3, 0.3, size=(n_runs, 3))
    seeds = np.arange(n_runs, dtype=np.uint32)
    kw = dict(n_points=200, lambda_=100.0, obs_factor=500.0, floating_base=1, use_momentum=1, use_hmc=1,
              hmc_resample_lambda=0.02)
    return goals, basegoals, seeds, kw


CONFIG5_KW = dict(n_points=200, lambda_=200.0, obs_factor=100.0)
CONFIG5_CUBE = 0.005
CONFIG5_PADDING = 0.15


def config5_goals(n_runs=4096):
    return np.random.default_rng(5).uniform(-0.8, 0.8, size=(n_runs, 30))


def config5_bodies():
    """the four box kinbodies of config 5: {name: ([(pose7, half3)], kinbody pose7)}"""
    return scenes.random_boxes(np.random.default_rng(20250104))


def config5_field_dims(boxes, cube_extent=CONFIG5_CUBE, padding=CONFIG5_PADDING):
    """grid of computedistancefield for a kinbody made of boxes at identity in its frame
    (reference src/orcdchomp_mod.cpp:377-409): sizes, lengths, grid pose in the kinbody frame"""
    lo = np.min([np.asarray(p[:3]) - np.asarray(h) for p, h in boxes], axis=0)
    hi = np.max([np.asarray(p[:3]) + np.asarray(h) for p, h in boxes], axis=0)
    apos = 0.5 * (lo + hi)
    aext = hi - apos
    return grid_dims(list(apos), list(aext), cube_extent, padding)


def pose_compose_np(ab, bc):
    """cd_kin_pose_compose (reference src/libcd/kin.c:136-178) in numpy, for test inputs"""
    ax, ay, az, aw = ab[3:7]
    bx, by, bz, bw = bc[3:7]
    q = [aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz]
    x, y, z = bc[:3]
    qx, qy, qz, qw = ax, ay, az, aw
    px = x * (qx * qx - qy * qy - qz * qz + qw * qw) + 2 * y * (qx * qy - qz * qw) + 2 * z * (qx * qz + qy * qw)
    py = 2 * x * (qx * qy + qz * qw) + y * (-qx * qx + qy * qy - qz * qz + qw * qw) + 2 * z * (qy * qz - qx * qw)
    pz = 2 * x * (qx * qz - qy * qw) + 2 * y * (qy * qz + qx * qw) + z * (-qx * qx - qy * qy + qz * qz + qw * qw)
    return np.array([px + ab[0], py + ab[1], pz + ab[2]] + q)


def config5_occupancy(cube_extent=CONFIG5_CUBE, padding=CONFIG5_PADDING):
    """occupancy of the four fields through the product's HOST voxelizer (the stand-in for OpenRAVE's
    collision checker; no GPU needed): [(name, occupancy, lengths, grid pose in the body frame, body pose)]"""
    import ctypes as C
    from or_cdchomp_amd import _capi
    lib = _capi.lib()
    bodies = config5_bodies()
    world_poses, halfs = [], []
    for name, (boxes, bpose) in bodies.items():
        for p, h in boxes:
            world_poses.append(pose_compose_np(np.asarray(bpose, dtype=np.float64), np.asarray(p, dtype=np.float64)))
            halfs.append(h)
    wp = np.ascontiguousarray(world_poses, dtype=np.float64)
    hf = np.ascontiguousarray(halfs, dtype=np.float64)
    out = []
    for name, (boxes, bpose) in bodies.items():
        sizes, lengths, gpose = config5_field_dims(boxes, cube_extent, padding)
        pw = pose_compose_np(np.asarray(bpose, dtype=np.float64), np.asarray(gpose, dtype=np.float64))
        occ = np.zeros(sizes)
        rc = lib.orc_host_voxelize_boxes(np.asarray(sizes, dtype=np.int32).ctypes.data_as(_capi.c_int_p),
                                         np.asarray(lengths, dtype=np.float64).ctypes.data_as(_capi.c_double_p),
                                         np.ascontiguousarray(pw).ctypes.data_as(_capi.c_double_p), C.c_double(cube_extent),
                                         len(wp), wp.ctypes.data_as(_capi.c_double_p), hf.ctypes.data_as(_capi.c_double_p),
                                         occ.ctypes.data_as(_capi.c_double_p))
        assert rc == 0
        out.append((name, occ, lengths, gpose, bpose))
    return out


def config5_oracle_fields(oracle_py, cube_extent=CONFIG5_CUBE, padding=CONFIG5_PADDING):
    """the four fields by the oracle (flood fill + signed distance transform of the occupancy above)"""
    grids, poses, names = [], [], []
    for name, occ, lengths, gpose, bpose in config5_occupancy(cube_extent, padding):
        g = oracle_py.OraGrid(occ, lengths)
        g.flood_fill(0)
        g.data[g.data == 1.0] = HUGE
        grids.append(g.bin_sdf())
        out = np.zeros(7)
        oracle_py.lib().ora_kin_pose_compose(oracle_py.dp(oracle_py.f64(bpose)), oracle_py.dp(oracle_py.f64(gpose)),
                                             oracle_py.dp(out))
        poses.append(out)
        names.append(name)
    return names, grids, poses


def setup_product_tree30(mod, cube_extent=CONFIG5_CUBE, padding=CONFIG5_PADDING):
    model = robots.tree30()
    mod.add_robot(model, transform=[0.0] * 6 + [1.0], dof_values=np.zeros(model.n_dof), active_dofs=list(range(model.n_dof)))
    for name, (boxes, pose) in config5_bodies().items():
        mod.add_kinbody_boxes(name, boxes, transform=pose)
    for name in config5_bodies():
        mod.SendCommand("computedistancefield kinbody %s cube_extent %f aabb_padding %f" % (name, cube_extent, padding))
    return model


# ---- the WAM of config 2 holding a four-sphere box (reference src/orcdchomp_mod.cpp:2168-2300; bench.py `held4`,
# tests/test_gpu_grabbed.py): 15 + 4 = 19 active spheres, the 32-lane kernel family in fp64 ----
HELD4_POS = [[0.0, 0.0, 0.0], [0.09, 0.0, 0.0], [0.0, 0.09, 0.02], [0.09, 0.09, 0.02]]
HELD4_RAD = [0.05, 0.045, 0.04, 0.05]
HELD4_OFFSET = (-0.04, -0.05, 0.15)      # of the box's frame from the handbase frame's origin, along its axes


def held4_pose(model):
    from or_cdchomp_amd import robots
    _, base, dofvals, _ = wam_state()
    R, t = model.link_frames(base, dofvals)
    li = model.link_names.index("handbase")
    return list(t[li] + R[li] @ np.asarray(HELD4_OFFSET)) + list(robots.quat_from_axis_angle((0.3, -0.5, 0.8), 0.7))


def setup_product_wam_held4(mod):
    """the config-2 scene with the box in the WAM's hand; returns (model, hand link index, box pose)"""
    model = setup_product_wam(mod)
    pose = held4_pose(model)
    hand = model.link_names.index("handbase")
    mod.add_kinbody_boxes("held4", [([0.045, 0.045, 0.01, 0, 0, 0, 1], [0.08, 0.08, 0.04])], transform=pose)
    mod.set_kinbody_spheres("held4", HELD4_POS, HELD4_RAD)
    mod.grab(model.name, "held4", hand)
    return model, hand, pose


def plan_switches_active():
    """an ORC_* experiment switch is set (scripts/test_toggles.sh runs the suite under each of them): the tests' assertions about WHICH
    kernel family or solve mode a batch was planned with do not apply then -- the answers must still be right"""
    import os
    return any(k.startswith("ORC_") and k not in ("ORC_RANDOM_ROBOTS", "ORC_LIB", "ORC_PHASE_TIMERS", "ORC_DEBUG_PLAN") for k in os.environ)
