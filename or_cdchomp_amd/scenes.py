"""Synthetic scenes: the table/mug meshes of the reference demo
(scripts/test_wam7.py:23,28) are OpenRAVE data files that are not in the repo, so
the configs use box stand-ins with the same role."""
import numpy as np

IDENT = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]


def tabletop_boxes():
    """table top 1.2 x 0.8 x 0.04 m with its top face at z=0.70 and a 0.08 x 0.08 x 0.12 m
    'mug' standing on it; returns {kinbody name: [(pose7, half_extents3), ...]}"""
    return {
        "table": [([0.0, 0.0, 0.68, 0, 0, 0, 1], [0.6, 0.4, 0.02])],
        "mug": [([0.2, 0.1, 0.76, 0, 0, 0, 1], [0.04, 0.04, 0.06])],
    }


def add_tabletop(mod):
    for name, boxes in tabletop_boxes().items():
        mod.add_kinbody_boxes(name, boxes, transform=IDENT)


def random_boxes(rng, n_bodies=4):
    """config 5: four box kinbodies with random poses (seeded by the caller)."""
    out = {}
    for k in range(n_bodies):
        half = rng.uniform(0.05, 0.2, size=3)
        pos = rng.uniform([-0.6, -0.6, 0.2], [0.6, 0.6, 1.4])
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        out["box%d" % k] = ([([0, 0, 0, 0, 0, 0, 1], list(half))], list(pos) + list(q))
    return out
