"""-m gpu: `computedistancefield` on random scenes (sizing, voxels, flood fill, signed distance
transform; src/orcdchomp_mod.cpp:377-409, 451-560, src/libcd/grid_flood.c:30-111, src/libcd/grid.c:462-687): the
device path and the host path give the same bits, and both the bits of the oracle's flood fill + `bin_sdf` on the same
occupancy -- for bodies of several boxes at any pose, hollow ones included (a closed shell's inside is not reached by
the flood from the corner and counts as inside the obstacle, as in the reference)."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from or_cdchomp_amd import _capi

pytestmark = pytest.mark.gpu
HUGE = float("inf")
SEEDS = list(range(int(os.environ.get("ORC_RANDOM_SCENES", "10"))))


def _random_quat(rng):
    q = rng.normal(size=4)
    return list(q / np.linalg.norm(q))


def _body(rng):
    """boxes [(pose7 in the body frame, half extents)] axis-aligned in the body frame, and the body's pose"""
    kind = int(rng.integers(0, 3))
    boxes = []
    if kind == 0:                                  # a few separate boxes
        for _ in range(int(rng.integers(1, 4))):
            boxes.append((list(rng.uniform(-0.25, 0.25, size=3)) + [0, 0, 0, 1], list(rng.uniform(0.02, 0.15, size=3))))
    elif kind == 1:                                # a closed shell: six slabs around a cavity
        a, b, c = rng.uniform(0.12, 0.25, size=3)
        w = float(rng.uniform(0.025, 0.05))
        for ax, half in ((0, [w, b, c]), (1, [a, w, c]), (2, [a, b, w])):
            for sgn in (-1.0, 1.0):
                pos = [0.0, 0.0, 0.0]; pos[ax] = sgn * ([a, b, c][ax] - w)
                boxes.append((pos + [0, 0, 0, 1], half))
    else:                                          # a shell with a window: the flood gets inside
        a, b, c = rng.uniform(0.12, 0.25, size=3)
        w = float(rng.uniform(0.025, 0.05))
        for ax, half in ((0, [w, b, c]), (1, [a, w, c]), (2, [a, b, w])):
            for sgn in (-1.0, 1.0):
                if ax == 2 and sgn > 0:
                    continue
                pos = [0.0, 0.0, 0.0]; pos[ax] = sgn * ([a, b, c][ax] - w)
                boxes.append((pos + [0, 0, 0, 1], half))
    pose = list(rng.uniform([-0.5, -0.5, 0.2], [0.5, 0.5, 1.2])) + _random_quat(rng)
    return boxes, pose, kind


@pytest.mark.parametrize("seed", SEEDS)
def test_random_scene_fields_device_host_oracle(oracle, monkeypatch, seed):
    import or_cdchomp_amd
    rng = np.random.default_rng(31000 + seed)
    boxes, pose, kind = _body(rng)
    cube = float(rng.uniform(0.008, 0.03))
    pad = float(rng.uniform(0.04, 0.25))
    got = {}
    for dev in ("1", "0"):
        monkeypatch.setenv("ORC_SDF_DEVICE", dev)
        mod = or_cdchomp_amd.Module(0)
        mod.add_kinbody_boxes("body", boxes, transform=pose)
        mod.SendCommand("computedistancefield kinbody body cube_extent %.17g aabb_padding %.17g" % (cube, pad))
        got[dev] = mod.get_sdf("body")
    (d1, l1, p1), (d0, l0, p0) = got["1"], got["0"]
    assert d1.shape == d0.shape and np.array_equal(l1, l0) and np.array_equal(p1, p0)
    assert np.array_equal(d1, d0)
    # the oracle on the occupancy of the host voxelizer (OpenRAVE's collision checker is third-party: the voxels are the build's own)
    sizes, lengths, gpose = common.config5_field_dims(boxes, cube, pad)
    assert list(d1.shape) == list(sizes) and np.allclose(l1, lengths, rtol=0, atol=0) and np.array_equal(np.asarray(gpose), p1)
    wp = np.ascontiguousarray([common.pose_compose_np(np.asarray(pose, dtype=np.float64), np.asarray(p, dtype=np.float64)) for p, h in boxes])
    hf = np.ascontiguousarray([h for p, h in boxes], dtype=np.float64)
    pw = np.ascontiguousarray(common.pose_compose_np(np.asarray(pose, dtype=np.float64), np.asarray(gpose, dtype=np.float64)))
    occ = np.zeros(sizes)
    lib = _capi.lib()
    assert lib.orc_host_voxelize_boxes(np.asarray(sizes, dtype=np.int32).ctypes.data_as(_capi.c_int_p),
                                       np.asarray(lengths, dtype=np.float64).ctypes.data_as(_capi.c_double_p),
                                       pw.ctypes.data_as(_capi.c_double_p), C.c_double(cube), len(wp),
                                       wp.ctypes.data_as(_capi.c_double_p), hf.ctypes.data_as(_capi.c_double_p),
                                       occ.ctypes.data_as(_capi.c_double_p)) == 0
    g = oracle.OraGrid(occ, lengths)
    g.flood_fill(0)
    g.data[g.data == 1.0] = HUGE
    n_obstacle = int(np.isinf(g.data).sum())
    want = g.bin_sdf().data
    assert np.array_equal(d1, want)
    hollow = int((want < -1.5 * 2 * cube).sum())
    print("scene %d: kind %d, grid %s, cells of %.1f mm, %d obstacle cells, %d deeper than 1.5 cells inside" % (
        seed, kind, tuple(sizes), 2e3 * cube, n_obstacle, hollow))
