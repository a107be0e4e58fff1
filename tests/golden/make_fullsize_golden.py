"""Writes tests/golden/fullsize_config{3,4,5}.npz: oracle outputs for a fixed sample of the runs of
the BASELINE.json configurations at their FULL size, so that the -m gpu tests can hold
the HIP path to the oracle at full size without a long oracle run on the GPU box.

    python tests/golden/make_fullsize_golden.py          (CPU only; about a minute)

Per configuration: the sample's run indices inside the full batch, the oracle's trajectories, costs
and status after n_iter iterations, and `self_amp`: the relative L2 distance between that trajectory
and the oracle's own trajectory when the goal is moved by ONE ulp, up or down, whichever moves it more -- the
conditioning of the run (chaotic runs of the reference algorithm are held to common.CHAOS_FACTOR times their measured
amplification; profiles/r04_chaos_ratio.txt is the distribution that factor comes from).
The inputs are not stored: tests/common.py rebuilds them from the seeds (config 5's occupancy comes
from the product's host voxelizer, the stand-in for OpenRAVE's collision checker; the fields
themselves are the oracle's flood fill + distance transform of it, stored here bit-packed so that
the test can demand the product's fields bit for bit).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common                                    # noqa: E402
from oracle import oracle_py as O                # noqa: E402
from or_cdchomp_amd import robots                # noqa: E402

N_ITER = 100
ULP = 1.0 + 2.0 ** -52
ULP_DOWN = 1.0 - 2.0 ** -52


def sample_indices(n_runs, count):
    return np.unique(np.linspace(0, n_runs - 1, count).astype(np.int64))


def run_wam(goals, kw, basegoals=None, seeds=None):
    prob = common.tabletop_problem(O)
    model, base, dofvals, adofs = common.wam_state()
    rob = O.OraRobot(model)
    p = O.default_params(**kw)
    out = []
    for g in (goals, goals * ULP, goals * ULP_DOWN):
        out.append(O.batch_run(rob, base, dofvals, adofs, g, [prob["sdf"]], [prob["pose"]], p, N_ITER,
                               basegoals=basegoals, seeds=seeds))
    return out


def pack(name, idx, res, extra=None):
    (traj, costs, status, _), (ptraj, _, pstatus, _), (mtraj, _, mstatus, _) = res
    amp = np.array([max(common.rel_l2(ptraj[k], traj[k]), common.rel_l2(mtraj[k], traj[k])) for k in range(len(idx))])
    d = dict(index=idx, traj=traj, costs=costs, status=status, self_amp=amp, status_goal_plus_one_ulp=pstatus,
             status_goal_minus_one_ulp=mstatus, n_iter=np.int64(N_ITER))
    if extra:
        d.update(extra)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **d)
    print("%s: %d runs, status %s, self_amp max %.2e, %d bytes" % (name, len(idx), np.unique(status).tolist(), amp.max(),
                                                                  os.path.getsize(path)))


def main():
    O.build(ref=False)
    # config 3: the first GPU's block (8192 runs) of the 65 536-run batch
    goals = common.config3_goals(rank=0, world=8)
    idx = sample_indices(len(goals), 64)
    pack("fullsize_config3.npz", idx, run_wam(goals[idx], common.CONFIG2_KW))

    # config 4: floating base + arm, momentum + hmc, batch 4096, seed = run index
    goals, basegoals, seeds, kw = common.config4_problem(4096)
    idx = sample_indices(4096, 32)
    pack("fullsize_config4.npz", idx, run_wam(goals[idx], kw, basegoals=basegoals[idx], seeds=seeds[idx]))

    # config 5: 30-dof tree, four fields at 1 cm cells, batch 4096 (the oracle computes in fp64)
    names, grids, poses = common.config5_oracle_fields(O)
    model = robots.tree30()
    rob = O.OraRobot(model)
    base = [0.0] * 6 + [1.0]
    dofvals = np.zeros(model.n_dof)
    adofs = list(range(model.n_dof))
    goals = common.config5_goals(4096)
    idx = sample_indices(4096, 32)
    p = O.default_params(**common.CONFIG5_KW)
    res = [O.batch_run(rob, base, dofvals, adofs, g, grids, poses, p, N_ITER) for g in (goals[idx], goals[idx] * ULP, goals[idx] * ULP_DOWN)]
    extra = {}
    for k, (name, occ, lengths, gpose, bpose) in enumerate(common.config5_occupancy()):
        extra["occ_bits_%d" % k] = np.packbits(np.isinf(occ).ravel())
        extra["occ_shape_%d" % k] = np.asarray(occ.shape, dtype=np.int64)
        # a checksum of the oracle's field: sum and sum of squares of the finite cells in a fixed order
        f = grids[k].data.ravel()
        extra["sdf_checksum_%d" % k] = np.array([f.sum(), np.square(f).sum(), f.min(), f.max()])
    pack("fullsize_config5.npz", idx, res, extra)


if __name__ == "__main__":
    main()
