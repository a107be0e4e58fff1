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


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
