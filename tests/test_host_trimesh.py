"""The host voxelizer for kinbodies given as triangle meshes (csrc/vox_tri.h; the collision queries of src/orcdchomp_mod.cpp:
462-531 for the reference's own kind of scene, scripts/test_wam7.py:23-28) against a linear program: a cube and a triangle
meet exactly when some point is a convex combination of the triangle's vertices AND inside the cube's six faces.  The product
decides with the 13-axis separating-axis test; scipy's HiGHS knows nothing of that.  CPU only (orc_host_voxelize_trimesh is
host code; the device kernel calls the same function and is compared with it bit for bit in tests/test_gpu_trimesh.py)."""
import ctypes as C

import numpy as np
import pytest
from scipy.optimize import linprog

from or_cdchomp_amd import _capi


def _rot(q):
    x, y, z, w = q
    return np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                     [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                     [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])


def _gap(c, R, h, tri):
    """how far the cube (centre c, axes R, half-extent h) would have to grow to reach the triangle: the least t with a point
    of the triangle inside the cube grown by t (t < 0: they overlap by -t, t > 0: they are t apart in the cube's max-norm)"""
    # variables: barycentric a, b, g, and t;  minimise t  s.t.  |R^T (a v0 + b v1 + g v2 - c)|_k <= h + t,  a + b + g = 1, a, b, g >= 0
    V = (R.T @ (np.asarray(tri) - c).T)                       # [3 axes][3 vertices] in the cube's frame
    A, b = [], []
    for k in range(3):
        A.append(list(V[k]) + [-1.0]); b.append(h)
        A.append(list(-V[k]) + [-1.0]); b.append(h)
    res = linprog(c=[0, 0, 0, 1.0], A_ub=np.array(A), b_ub=np.array(b), A_eq=[[1, 1, 1, 0]], b_eq=[1.0],
                  bounds=[(0, None)] * 3 + [(None, None)], method="highs")
    assert res.status == 0, res.message
    return res.x[3]


def _voxelize(sizes, lengths, gpose, cube, tris):
    lib = _capi.lib()
    occ = np.zeros(sizes)
    t = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 9)
    assert lib.orc_host_voxelize_trimesh(np.asarray(sizes, dtype=np.int32).ctypes.data_as(_capi.c_int_p),
                                         np.asarray(lengths, dtype=np.float64).ctypes.data_as(_capi.c_double_p),
                                         np.ascontiguousarray(gpose, dtype=np.float64).ctypes.data_as(_capi.c_double_p), C.c_double(cube), len(t),
                                         t.ctypes.data_as(_capi.c_double_p), occ.ctypes.data_as(_capi.c_double_p)) == 0
    return occ


@pytest.mark.parametrize("seed", range(6))
def test_voxels_are_the_cells_whose_cube_meets_a_triangle(seed):
    rng = np.random.default_rng(71000 + seed)
    n_tri = int(rng.integers(2, 9))
    tris = []
    for _ in range(n_tri):
        centre = rng.uniform(-0.12, 0.12, size=3)
        tris.append(centre + rng.normal(scale=rng.uniform(0.02, 0.15), size=(3, 3)))
    if seed == 5:
        tris.append(np.array([[0.0, 0.0, 0.0], [0.1, 0.0, 0.0], [0.2, 0.0, 0.0]]))        # a degenerate triangle (a segment): no normal
    cube = float(rng.uniform(0.01, 0.03))
    sizes = [int(v) for v in rng.integers(8, 15, size=3)]
    lengths = [s * 2 * cube for s in sizes]
    gq = rng.normal(size=4); gq /= np.linalg.norm(gq)
    Rg = _rot(gq)
    gpose = np.array(list(-Rg @ (0.5 * np.array(lengths))) + list(gq))
    occ = _voxelize(sizes, lengths, gpose, cube, tris)
    assert 0 < np.isinf(occ).sum() < occ.size and set(np.unique(occ[~np.isinf(occ)])) == {1.0}
    inside = np.isinf(occ)
    edge = np.zeros_like(inside)
    for ax in range(3):
        edge |= inside != np.roll(inside, 1, axis=ax)
        edge |= inside != np.roll(inside, -1, axis=ax)
    cells = np.argwhere(edge)
    cells = cells[rng.permutation(len(cells))[:200]]
    cells = np.vstack([cells, np.column_stack([rng.integers(0, s, size=40) for s in sizes])])
    checked = 0
    for ijk in cells:
        cw = Rg @ ((ijk + 0.5) * 2 * cube) + gpose[:3]
        t = min(_gap(cw, Rg, cube, tri) for tri in tris)
        if abs(t) < 1e-6:
            continue                                            # touching to within the tolerance of either method
        assert bool(np.isinf(occ[tuple(ijk)])) == (t < 0), (seed, ijk, t)
        checked += 1
    assert checked > 180


def test_touching_counts_and_a_closed_mesh_gives_a_closed_shell():
    """a box given as 12 triangles whose faces lie exactly on cell boundaries: the cells either side of a face touch it and both
    are obstacle cells, so the flood fill from the corner (src/orcdchomp_mod.cpp:540-548) cannot leak into the box"""
    lib = _capi.lib()
    cube = 0.02
    sizes = [12, 10, 8]
    lengths = [s * 2 * cube for s in sizes]
    gpose = [0, 0, 0, 0, 0, 0, 1.0]
    lo = np.array([0.16, 0.12, 0.08]); hi = np.array([0.32, 0.28, 0.24])          # faces on multiples of 0.04
    occ = _voxelize(sizes, lengths, gpose, cube, _box_triangles(lo, hi))
    cells = occ.copy()
    assert lib.orc_host_flood_fill(np.asarray(sizes, dtype=np.int32).ctypes.data_as(_capi.c_int_p), cells.ctypes.data_as(_capi.c_double_p), 0) == 0
    obstacle = cells != 0.0                                                          # not reached from the corner
    idx = np.indices(sizes).transpose(1, 2, 3, 0)
    centre = (idx + 0.5) * 2 * cube
    box_cells = np.all((centre > lo) & (centre < hi), axis=3)
    grown = np.all((centre > lo - 2 * cube) & (centre < hi + 2 * cube), axis=3)       # one layer of touching cells around it
    assert obstacle[box_cells].all()                                                 # the inside is closed off
    assert np.array_equal(obstacle, grown)                                           # and the shell is exactly the touching layer


def _box_triangles(lo, hi):
    x0, y0, z0 = lo; x1, y1, z1 = hi
    c = [np.array(p, dtype=float) for p in ((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))]
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (2, 3, 7, 6), (1, 2, 6, 5), (3, 0, 4, 7)]      # outward winding
    tris = []
    for a, b, cc, d in quads:
        tris.append([c[a], c[b], c[cc]]); tris.append([c[a], c[cc], c[d]])
    return np.array(tris)
