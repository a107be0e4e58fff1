"""not gpu: the oracle against the reference's own compiled code (oracle/_ref/libcd_ref.so), live,
on randomised inputs beyond the committed golden vectors.  Skipped when _ref is absent."""
import ctypes as C

import numpy as np
import pytest


class CdGrid(C.Structure):          # struct cd_grid, /root/reference src/libcd/grid.h:29-41
    _fields_ = [("n", C.c_int), ("sizes", C.POINTER(C.c_int)), ("ncells", C.c_size_t),
                ("cell_size", C.c_int), ("data", C.c_void_p), ("lengths", C.POINTER(C.c_double))]


@pytest.fixture(scope="module")
def ref(oracle):
    R = oracle.ref()
    if R is None:
        pytest.skip("oracle/_ref/libcd_ref.so not built (needs /root/reference)")
    R.cd_grid_double_interp.argtypes = [C.POINTER(CdGrid), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    R.cd_grid_double_grad.argtypes = [C.POINTER(CdGrid), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    R.cd_grid_lookup_index.argtypes = [C.POINTER(CdGrid), C.POINTER(C.c_double), C.POINTER(C.c_size_t)]
    return R


def _ref_grid(R, data, lengths):
    data = np.ascontiguousarray(data, dtype=np.float64)
    sizes = (C.c_int * 3)(*data.shape)
    gp = C.POINTER(CdGrid)()
    init = C.c_double(0.0)
    R.cd_grid_create_sizearray(C.byref(gp), C.byref(init), 8, 3, sizes)
    C.memmove(gp.contents.data, data.ctypes.data, data.nbytes)
    for i in range(3):
        gp.contents.lengths[i] = lengths[i]
    return gp


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_sdf_pipeline_random_scenes(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    shape = tuple(int(v) for v in rng.integers(5, 20, size=3))
    lengths = rng.uniform(0.3, 2.0, size=3)
    occ = np.where(rng.uniform(size=shape) < 0.06, np.inf, 0.0)
    occ.flat[0] = 0.0
    gp = _ref_grid(ref, occ, lengths)
    out = C.POINTER(CdGrid)()
    assert ref.cd_grid_double_bin_sdf(C.byref(out), gp) == 0
    want = np.ctypeslib.as_array(C.cast(out.contents.data, C.POINTER(C.c_double)), shape=shape).copy()
    og = oracle.OraGrid(occ, lengths)
    sdf = og.bin_sdf()
    assert np.array_equal(sdf.data, want)
    pts = rng.uniform(-0.05, 1.05, size=(400, 3)) * lengths
    for p in pts:
        p = np.ascontiguousarray(p)
        v = C.c_double(); g = np.zeros(3); idx = C.c_size_t()
        e = ref.cd_grid_double_interp(out, p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(v))
        oe, ov = sdf.interp(p)
        assert e == oe
        if e == 0:
            assert v.value == ov
            ref.cd_grid_double_grad(out, p.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)))
            _, og_ = sdf.grad(p)
            assert np.array_equal(g, og_)
            oidx = C.c_size_t()
            ref.cd_grid_lookup_index(out, p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(idx))
            oracle.lib().ora_grid_lookup_index(sdf.ptr, oracle.dp(p), C.byref(oidx))
            assert idx.value == oidx.value
