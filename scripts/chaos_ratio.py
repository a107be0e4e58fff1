"""How far may the HIP path be from the oracle on a run the oracle itself cannot reproduce?

For every run of a sample of config 2 and config 4: err = rel L2 (HIP vs oracle), and the oracle's own movement under
one-ulp changes of the goal (amp_a: goal x (1 + 2^-52), the experiment the tests make; amp_b: the larger of that and
goal x (1 - 2^-52); amp_c: the largest of those and one ulp on the first joint alone).  Prints the distribution of
err / amp over the ill-conditioned runs (amp >= 1e-9) and the status agreement, which is what tests/common.py's
CHAOS_FACTOR and the status bars of the tests rest on.  Output kept as profiles/r04_chaos_ratio.txt."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
import or_cdchomp_amd
from oracle import oracle_py as O

O.build(ref=False)
N_ITER = 100


def study(tag, mod, model, goals, kw, ora_args, fields, extra=None):
    extra = extra or {}
    bid = mod.batch_create(model.name, goals, **dict(kw, **{k: v for k, v in extra.items()}))
    costs, status = mod.batch_iterate(bid, N_ITER)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    okw = dict(kw)
    p = O.default_params(**okw)
    bkw = {k: v for k, v in extra.items()}

    def ora(g):
        return O.batch_run(*ora_args, g, fields[0], fields[1], p, N_ITER, **bkw)[:3]
    ot, oc, ost = ora(goals)
    pert = [goals * (1 + 2.0 ** -52), goals * (1 - 2.0 ** -52)]
    g3 = goals.copy(); g3[:, 0] = np.nextafter(g3[:, 0], np.inf)
    pert.append(g3)
    amps, psts = [], []
    for g in pert:
        pt, _, pst = ora(g)
        amps.append(np.array([common.rel_l2(pt[k], ot[k]) for k in range(len(goals))]))
        psts.append(pst)
    amp_a = amps[0]; amp_b = np.maximum(amps[0], amps[1]); amp_c = np.maximum(amp_b, amps[2])
    err = np.array([common.rel_l2(traj[k], ot[k]) for k in range(len(goals))])
    both = (status == 0) & (ost == 0)
    print("== %s: %d runs, %d iterations" % (tag, len(goals), N_ITER))
    print("status: hip -1: %d, oracle -1: %d, agree %.4f; oracle vs oracle(goal + 1 ulp) agree %.4f, vs (goal - 1 ulp) %.4f"
          % ((status != 0).sum(), (ost != 0).sum(), (status == ost).mean(), (ost == psts[0]).mean(), (ost == psts[1]).mean()))
    dis = np.flatnonzero(status != ost)
    stable = [(psts[0][k] == ost[k]) and (psts[1][k] == ost[k]) and (psts[2][k] == ost[k]) for k in dis]
    print("  runs whose status differs: %d; of these the oracle keeps its own status under all three one-ulp changes: %d  %s"
          % (len(dis), int(np.sum(stable)), [(int(k), float(amp_c[k])) for k, s in zip(dis, stable) if s][:8]))
    well = both & (amp_a < 1e-9)
    print("well-conditioned (amp_a < 1e-9, status 0 in both): %d runs, err max %.3e median %.3e" % (well.sum(), err[well].max(), np.median(err[well])))
    for name, amp in (("amp_a (+1 ulp)", amp_a), ("amp_b (+-1 ulp)", amp_b), ("amp_c (three changes)", amp_c)):
        ill = both & (amp >= 1e-9)
        if not ill.any():
            print("  %s: no ill-conditioned run" % name); continue
        r = err[ill] / amp[ill]
        q = np.percentile(r, [0, 50, 90, 99, 100])
        print("  %-22s ill-conditioned %4d runs: err/amp min %.2e median %.2e p90 %.2e p99 %.2e max %.2e ; err max %.2e ; runs above 1e-6 with ratio > 10: %d, > 50: %d"
              % (name, ill.sum(), q[0], q[1], q[2], q[3], q[4], err[ill].max(), int(((r > 10) & (err[ill] > 1e-6)).sum()), int(((r > 50) & (err[ill] > 1e-6)).sum())))
    # runs well-conditioned by amp_a that still miss 1e-6 would be defects
    miss = both & (amp_c < 1e-9) & (err > 1e-6)
    print("  runs with amp_c < 1e-9 and err > 1e-6 (would be defects): %d" % miss.sum())
    sys.stdout.flush()


def main():
    n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    n4 = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    prob = common.tabletop_problem(O)
    _, base, dofvals, adofs = common.wam_state()
    rob = O.OraRobot(model)
    goals = np.concatenate([common.wam_goals(1024, seed=20250101 + s) for s in range((n2 + 1023) // 1024)])[:n2]
    study("config 2", mod, model, goals, dict(common.CONFIG2_KW), (rob, base, dofvals, adofs), ([prob["sdf"]], [prob["pose"]]))
    g4, bg4, sd4, kw4 = common.config4_problem(4096)
    study("config 4", mod, model, g4[:n4], kw4, (rob, base, dofvals, adofs), ([prob["sdf"]], [prob["pose"]]),
          extra=dict(basegoals=bg4[:n4], seeds=sd4[:n4]))


if __name__ == "__main__":
    main()
