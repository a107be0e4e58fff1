"""How many sphere pairs of the WAM are within range of the self-collision term, per waypoint and per wavefront pass
(4 waypoints x 16 lanes), on the straight-line trajectories of BASELINE config 2 -- the numbers behind the question
"rotations taken by the whole wavefront" against "rounds of each lane's own pairs".   python scripts/self_pair_stats.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import common
from oracle import oracle_py
model, base, dofvals, adofs = common.wam_state()
rob = oracle_py.OraRobot(model)
a = model.arrays()
link = np.asarray(a["sphere_link"]); pos = np.asarray(a["sphere_pos"]).reshape(-1, 3); rad = np.asarray(a["sphere_radius"])
eps = float(oracle_py.default_params().epsilon_self)
# active spheres: on links some active dof moves (link index > 0 of the arm); the product's slot placement for them (ORC_DEBUG_PLAN)
slots = [int(x) for x in os.environ.get("SLOTS", "14 0 1 6 4 2 5 3 15 10 12 11 9 7 8 13").split()]
goals = common.wam_goals(64)
n_points = 100
R0, t0, _, _ = rob.fk(base, dofvals)
moved = np.zeros(len(link), bool)
for g in goals[:4]:
    q = dofvals.copy(); q[:7] = g
    R1, t1, _, _ = rob.fk(base, q)
    moved |= np.array([not (np.allclose(R0[l], R1[l]) and np.allclose(t0[l], t1[l])) for l in link])
act = np.concatenate([np.nonzero(moved)[0], np.nonzero(~moved)[0]])      # the inactive spheres ride on the free lanes, after the active ones
print("spheres %d, active %d, eps_self %.3f" % (len(link), len(act), eps))
assert len(act) == len(slots), (len(act), len(slots))
slot = np.array(slots)
masks = []      # per waypoint: [16 lanes] bitmask over K = 1..8 of near pairs the lane owns (K = 8: both lanes)
npairs = []
for g in goals:
    for i in range(1, n_points - 1):
        q = dofvals.copy(); q[:7] = dofvals[:7] + (g - dofvals[:7]) * i / (n_points - 1)
        R, t, _, _ = rob.fk(base, q)
        pw = np.einsum("sij,sj->si", R[link[act]], pos[act]) + t[link[act]]
        m = np.zeros(16, dtype=int); cnt = 0
        for x in range(len(act)):
            for y in range(x + 1, len(act)):
                if link[act[x]] == link[act[y]]: continue
                Rr = rad[act[x]] + rad[act[y]] + eps
                if ((pw[x] - pw[y]) ** 2).sum() <= Rr * Rr:
                    cnt += 1
                    sx, sy = slot[x], slot[y]
                    K = (sy - sx) % 16
                    if K <= 8: m[sx] |= 1 << K
                    if K >= 8: m[sy] |= 1 << (16 - K)
        masks.append(m); npairs.append(cnt)
masks = np.array(masks); npairs = np.array(npairs)
print("pairs in range per waypoint: mean %.1f  min %d  max %d  (of %d candidate pairs)" % (npairs.mean(), npairs.min(), npairs.max(), sum(1 for x in range(len(act)) for y in range(x+1, len(act)) if link[act[x]] != link[act[y]])))
pc = np.array([[bin(v).count("1") for v in m] for m in masks])
w = len(masks) // 4 * 4
un = np.bitwise_or.reduce(masks[:w].reshape(-1, 4 * 16), axis=1)
print("per wavefront pass (4 consecutive waypoints): rotations some lane takes (the shipped form) mean %.2f ; the most pairs ONE lane owns mean %.2f  max %d ; pairs in the wavefront mean %.1f -> %.2f rounds of 64 if they were compacted" % (
    np.mean([bin(v).count("1") for v in un]), pc[:w].reshape(-1, 64).max(axis=1).mean(), pc.max(), 4 * npairs.mean(), np.mean(np.ceil(npairs[:w].reshape(-1, 4).sum(axis=1) / 64.0))))
print("histogram of the per-wavefront maximum of pairs per lane:", np.bincount(pc[:w].reshape(-1, 64).max(axis=1)))
