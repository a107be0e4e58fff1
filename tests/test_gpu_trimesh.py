"""-m gpu: computedistancefield over kinbodies given as triangle meshes (orc_env_add_kinbody_trimesh; the collision queries of
src/orcdchomp_mod.cpp:462-531 for the reference's own kind of scene, scripts/test_wam7.py:23-28: rolly-table.iv, mug3.iv).

* the tabletop given as 12 triangles per box produces the field of the box path bit for bit (a mesh is a surface: its inside is
  what the flood fill of src/orcdchomp_mod.cpp:540-548 cannot reach);
* the device kernel (csrc/sdf_kernels.hip) and the host path call the same function (csrc/vox_tri.h): the same cells bit for bit,
  on random closed and open meshes;
* the cache file is what it was (raw doubles, validated by size)."""
import os

import numpy as np
import pytest

import common
import or_cdchomp_amd
from or_cdchomp_amd import robots, scenes

pytestmark = pytest.mark.gpu


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                     [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                     [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])


def box_triangles(pose, half):
    """the 12 triangles (outward winding) of an oriented box: pose [x y z qx qy qz qw] in the kinbody frame"""
    R = _rot(pose[3:]); t = np.asarray(pose[:3], dtype=float)
    sg = [(-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1), (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1)]
    c = [R @ (np.array(s, dtype=float) * np.asarray(half)) + t for s in sg]
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (3, 0, 4, 7)]
    tris = []
    for a, b, cc, d in quads:
        tris.append([c[a], c[b], c[cc]]); tris.append([c[a], c[cc], c[d]])
    return np.array(tris)


def _tabletop(mod, as_mesh, transform, shrink=1.0):
    """the table and the mug of BASELINE configs[0] (scenes.tabletop_boxes) as boxes, or every box as 12 triangles"""
    for name, boxes in scenes.tabletop_boxes().items():
        boxes = [(p, [shrink * v for v in h]) for p, h in boxes]
        if as_mesh:
            tris = np.concatenate([box_triangles(np.asarray(p, dtype=float), h) for p, h in boxes])
            mod.add_kinbody_trimesh(name, tris, transform=transform if transform is not None else scenes.IDENT)
        else:
            mod.add_kinbody_boxes(name, boxes, transform=transform if transform is not None else scenes.IDENT)


@pytest.mark.parametrize("device_build", [0, 1])
def test_the_tabletop_as_triangles_is_the_tabletop_as_boxes(device_build, monkeypatch):
    """BASELINE configs[0]'s table and mug, each box as 12 triangles: the same field bit for bit, on the host path and on the device
    path.  The boxes are 1.7 % smaller than in scenes.py: those have every face exactly on a cell boundary (0.6 / 0.4 / 0.02 m half
    extents, 4 cm cells, and the field's frame is the kinbody's, so no body pose changes that), where a mesh -- for which touching
    has to count, or a closed mesh would not give a closed shell -- is one layer of touching cells fatter than a box
    (tests/test_host_trimesh.py::test_touching_counts_and_a_closed_mesh_gives_a_closed_shell)"""
    monkeypatch.setenv("ORC_SDF_DEVICE", str(device_build))
    pose = [0.0137, -0.0071, 0.0093] + list(robots.quat_from_axis_angle((0.2, -0.1, 1.0), 0.31))
    fields = []
    for as_mesh in (False, True):
        mod = or_cdchomp_amd.Module(0)
        _tabletop(mod, as_mesh, pose, shrink=0.983)
        mod.SendCommand("computedistancefield kinbody table")
        fields.append(mod.get_sdf("table"))
        mod.close()
    (a, la, pa), (b, lb, pb) = fields
    assert a.shape == b.shape and np.array_equal(la, lb) and np.array_equal(pa, pb)      # the same grid: sized from the same box around the geometry
    assert np.array_equal(a, b)
    assert (a < 0).any() and (a > 0).any()


def test_device_and_host_agree_on_random_meshes(tmp_path, monkeypatch):
    """closed shells (boxes as triangles, rotated) plus loose triangles that stick out of them, any body pose, 16-40 mm cells: the
    device build, the host build and a cache file written by one and read by the other are the same doubles"""
    rng = np.random.default_rng(8101)
    for case in range(6):
        tris = []
        for _ in range(int(rng.integers(1, 4))):
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            tris.append(box_triangles(np.array(list(rng.uniform(-0.2, 0.2, size=3)) + list(q)), rng.uniform(0.04, 0.15, size=3)))
        loose = rng.uniform(-0.3, 0.3, size=(int(rng.integers(0, 5)), 1, 3)) + rng.normal(scale=0.08, size=(1, 3, 3))
        tris = np.concatenate(tris + ([loose] if len(loose) else []))
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pose = list(rng.uniform(-0.5, 0.5, size=3)) + list(q)
        cube = float(rng.uniform(0.008, 0.02))
        out = []
        cache = str(tmp_path / ("mesh%d.dat" % case))
        for dev in (1, 0):
            monkeypatch.setenv("ORC_SDF_DEVICE", str(dev))
            mod = or_cdchomp_amd.Module(0)
            mod.add_kinbody_trimesh("thing", tris, transform=pose)
            mod.SendCommand("computedistancefield kinbody thing cube_extent %r aabb_padding 0.1%s" % (cube, " cache_filename %s" % cache if dev == 1 else ""))
            out.append(mod.get_sdf("thing")[0])
            mod.close()
        assert np.array_equal(out[0], out[1]), case
        assert os.path.getsize(cache) == out[0].size * 8
        mod = or_cdchomp_amd.Module(0)
        mod.add_kinbody_trimesh("thing", tris, transform=pose)
        mod.SendCommand("computedistancefield kinbody thing cube_extent %r aabb_padding 0.1 cache_filename %s require_cache" % (cube, cache))
        assert np.array_equal(mod.get_sdf("thing")[0], out[0])
        mod.close()
        assert (out[0] < 0).any()                    # the shells have an inside


def test_a_mesh_kinbody_in_a_run_and_error_paths():
    """a run against the field of a mesh kinbody iterates like any other; the C ABI's argument checks"""
    mod = or_cdchomp_amd.Module(0)
    model, base, dofvals, adofs = common.wam_state()
    mod.add_robot(model, transform=base, dof_values=dofvals, active_dofs=adofs)
    _tabletop(mod, True, None)
    mod.SendCommand("computedistancefield kinbody table")
    goals = common.wam_goals(8, seed=3)
    bid = mod.batch_create(model.name, goals, **common.CONFIG2_KW)
    costs, status = mod.batch_iterate(bid, 20)
    assert np.isfinite(costs).all()
    mod.batch_destroy(bid)
    lib = mod._lib
    import ctypes as C
    v = np.zeros(9)
    assert lib.orc_env_add_kinbody_trimesh(mod._h, b"m", 0, v.ctypes.data_as(C.POINTER(C.c_double))) == 1
    assert lib.orc_env_add_kinbody_trimesh(mod._h, b"m", 1, None) == 1
    assert lib.orc_env_add_kinbody_trimesh(mod._h, None, 1, v.ctypes.data_as(C.POINTER(C.c_double))) == 1
    v[4] = np.nan
    assert lib.orc_env_add_kinbody_trimesh(mod._h, b"m", 1, v.ctypes.data_as(C.POINTER(C.c_double))) == 1
    assert lib.orc_env_add_kinbody_trimesh(mod._h, model.name.encode(), 1, np.zeros(9).ctypes.data_as(C.POINTER(C.c_double))) == 1      # a robot of that name
    mod.close()
