#!/usr/bin/env python3
"""Generates tests/golden/*.npz|json from the reference's OWN code (oracle/_ref/libcd_ref.so =
/root/reference/src/libcd/{grid,grid_flood,mat,util_shparse}.c compiled unmodified, see
oracle/Makefile).  Run in the build container only (needs /root/reference); the outputs are
data (inputs + expected outputs) and are committed.

   python tests/golden/make_golden.py
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle_py as O  # noqa: E402

O.build(ref=True)
R = O.ref()
assert R is not None, "oracle/_ref/libcd_ref.so missing (needs /root/reference)"


class CdGrid(C.Structure):          # struct cd_grid, src/libcd/grid.h:29-41
    _fields_ = [("n", C.c_int), ("sizes", C.POINTER(C.c_int)), ("ncells", C.c_size_t),
                ("cell_size", C.c_int), ("data", C.c_void_p), ("lengths", C.POINTER(C.c_double))]


def ref_grid(data, lengths):
    data = np.ascontiguousarray(data, dtype=np.float64)
    sizes = (C.c_int * 3)(*data.shape)
    gp = C.POINTER(CdGrid)()
    init = C.c_double(0.0)
    R.cd_grid_create_sizearray(C.byref(gp), C.byref(init), 8, 3, sizes)
    C.memmove(gp.contents.data, data.ctypes.data, data.nbytes)
    for i in range(3):
        gp.contents.lengths[i] = lengths[i]
    return gp


def ref_data(gp, shape):
    return np.ctypeslib.as_array(C.cast(gp.contents.data, C.POINTER(C.c_double)), shape=shape).copy()


rng = np.random.default_rng(1234)

# ---- (2) SDF: 24^3 synthetic scene -> cd_grid_double_bin_sdf, + interp/grad probes
shape = (24, 24, 24)
lengths = [0.96, 1.2, 0.72]
occ = np.zeros(shape)
occ[4:9, 5:20, 3:6] = np.inf          # a slab
occ[14:18, 8:12, 10:20] = np.inf      # a pillar
occ[20, 20, 20] = np.inf              # a single cell
g_occ = ref_grid(occ, lengths)
g_sdf = C.POINTER(CdGrid)()
assert R.cd_grid_double_bin_sdf(C.byref(g_sdf), g_occ) == 0
sdf = ref_data(g_sdf, shape)

# probes: inside, outside, exactly on borders, near HUGE_VAL cells (use the occupancy grid for those)
pts = rng.uniform(-0.1, 1.1, size=(256, 3)) * np.asarray(lengths)
pts[:8] = [[0, 0, 0], lengths, [0.5 * lengths[0], 0, lengths[2]], [lengths[0], 0.3, 0.2],
           [0.02, 0.02, 0.02], [0.94, 1.18, 0.70], [0.2, 0.5, 0.12], [-1e-12, 0.5, 0.3]]


def probe(gp, pts):
    vals = np.zeros(len(pts)); grads = np.zeros((len(pts), 3)); errs = np.zeros(len(pts), dtype=np.int32)
    R.cd_grid_double_interp.argtypes = [C.POINTER(CdGrid), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    R.cd_grid_double_grad.argtypes = [C.POINTER(CdGrid), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for i, p in enumerate(pts):
        p = np.ascontiguousarray(p)
        v = C.c_double(0.0)
        errs[i] = R.cd_grid_double_interp(gp, p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(v))
        vals[i] = v.value if errs[i] == 0 else 0.0
        g = np.zeros(3)
        if errs[i] == 0:
            R.cd_grid_double_grad(gp, p.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)))
        grads[i] = g
    return vals, grads, errs


sdf_vals, sdf_grads, sdf_errs = probe(g_sdf, pts)
# a grid with HUGE_VAL cells to exercise the poisoning rules of interp
poison = rng.uniform(0, 1, size=(6, 7, 8))
poison[2, 3, 4] = np.inf; poison[0, 0, 0] = np.inf; poison[5, 6, 7] = np.inf; poison[3, :, 2] = np.inf
plen = [0.6, 0.7, 0.8]
g_p = ref_grid(poison, plen)
ppts = rng.uniform(0, 1, size=(256, 3)) * np.asarray(plen)
p_vals, p_grads, p_errs = probe(g_p, ppts)
p_grads[~np.isfinite(p_grads)] = np.nan   # inf-inf in the reference's grad; compared as "not finite"

# flood fill (6-connected) with replace_1_to_0 semantics
CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_void_p)


@CB
def replace_1_to_0(val, rptr):
    if val[0] == 1.0:
        val[0] = 0.0
        return 1
    return 0


ff = np.ones((9, 8, 7))
ff[3, :, :] = np.inf          # a wall ...
ff[3, 4, 3] = 1.0             # ... with a hole
ff[5:8, 2:5, 2:5] = np.inf    # a closed box
ff[6, 3, 3] = 1.0             # free cell sealed inside -> must stay 1.0
ff_in = ff.copy()
g_ff = ref_grid(ff, [1, 1, 1])
R.cd_grid_flood_fill.argtypes = [C.POINTER(CdGrid), C.c_size_t, C.c_void_p, CB, C.c_void_p]
R.cd_grid_flood_fill(g_ff, 0, None, replace_1_to_0, None)
ff_out = ref_data(g_ff, ff.shape)

np.savez_compressed(os.path.join(HERE, "grid_golden.npz"),
                    occ=occ, lengths=np.asarray(lengths), sdf=sdf, pts=pts, sdf_vals=sdf_vals, sdf_grads=sdf_grads,
                    sdf_errs=sdf_errs, poison=poison, plen=np.asarray(plen), ppts=ppts, p_vals=p_vals,
                    p_grads=p_grads, p_errs=p_errs, ff_in=ff_in, ff_out=ff_out)

# ---- shparse cases through the reference tokenizer
cases = ["create robot 'BarrettWAM' adofgoal '0.6 -1.2 0.3' lambda 100.0000",
         "a  b\tc\n d", "'it'\\''s' \"dq \\\" x\" plain\\ space", "\"a\\\\b\" 'c\\d' e\\\nf",
         "  leading and trailing  ", "x\"y z\"w 'p q'r", "\"in dq \\n stays\" \\'"]
R.cd_util_shparse.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_char_p))]
out = []
for s in cases:
    buf = C.create_string_buffer(s.encode())
    argc = C.c_int(); argv = C.POINTER(C.c_char_p)()
    R.cd_util_shparse(buf, C.byref(argc), C.byref(argv))
    out.append({"in": s, "tokens": [argv[i].decode() for i in range(argc.value)]})
json.dump(out, open(os.path.join(HERE, "shparse_golden.json"), "w"), indent=1)

# ---- known answers recorded from the reference build during the survey;
# chomp.c cannot be compiled here (needs cblas/lapacke), these are the pins for its restatement
json.dump({
    "source": "SURVEY.md section 8(c), 'Probe known-answers obtained'",
    "chomp": {"n_points": 101, "n": 7, "D": 1, "dt": 0.01, "goal": "q_j = 0.3*(j+1), start 0",
              "A[0][0..2]": [200.0, -100.0, 0.0], "Ainv[0][0]": 0.0099, "Ainv[49][49]": 0.25,
              "Kvels[0][1]": 50.0, "Kvels[1][0]": -50.0, "B[98][0]": -30.0, "trC": 630.0,
              "smooth_cost_converged": 6.3, "T[50][3]_converged": 0.612},
    "grid": {"sizes": [4, 5, 6], "cell": 0.04, "obstacle": [2, 2, 3], "sdf[0]": 0.164924, "sdf[obs]": -0.04,
             "p": [0.05, 0.11, 0.13], "interp": 0.0582842712, "grad": [-1.0, 0.414214, -0.414214]},
}, open(os.path.join(HERE, "survey_known_answers.json"), "w"), indent=1)
print("golden fixtures written to", HERE)
