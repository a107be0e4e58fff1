import sys, os
os.environ["ORC_PHASE_TIMERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np, time
import common
import or_cdchomp_amd
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
goals = common.wam_goals(n_runs)
kw = dict(n_points=100, lambda_=100.0, obs_factor=500.0)
if len(sys.argv) > 2: kw["derivative"] = int(sys.argv[2])      # python scripts/phase_profile.py 1024 2
bid = mod.batch_create(model.name, goals, **kw)
mod.batch_iterate(bid, 5)
mod.kernel_time(reset=True)
t0 = time.time(); mod.batch_iterate(bid, 100); t1 = time.time()
ms, n = mod.kernel_time()
ph = mod._lib  # noqa
out = np.zeros((n_runs, 8))
import ctypes as C
mod._check(mod._lib.orc_batch_get_state(mod._h, bid, b"phase", out.ctypes.data_as(C.POINTER(C.c_double)), out.size))
names = ["FK", "cost", "obs-reduce", "smooth+solve+step", "joint limits", "smooth cost", "-", "-"]
tot = out[:, :6].sum(1)
print("runs %d  kernel %.2f ms  -> %.3g it/s ; mean cycles/iteration per WG %.0f" % (n_runs, ms, n_runs*100/(ms*1e-3), tot.mean()/101))
for k in range(6):
    print("  %-18s %8.0f cycles/iter  %5.1f %%" % (names[k], out[:, k].mean()/101, 100*out[:, k].sum()/tot.sum()))
tot_all = out[:, :6].sum(1)
q = np.percentile(tot_all, [0, 10, 50, 90, 99, 100]) / 1e6
print("per-WG total Mcycles: min %.1f p10 %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % tuple(q))
jl = out[:, 4]
print("joint-limit Mcycles: median %.2f p90 %.2f p99 %.2f max %.2f ; share of runs with >20%% of time there: %.1f %%" % (
    np.median(jl)/1e6, np.percentile(jl, 90)/1e6, np.percentile(jl, 99)/1e6, jl.max()/1e6, 100*np.mean(jl > 0.2*tot_all)))
rounds = out[:, 6]
print("limit rounds per run (100 iterations): mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (
    rounds.mean(), np.median(rounds), np.percentile(rounds, 90), np.percentile(rounds, 99), rounds.max()))
# kinds of joint-limit rounds (ph[7] packs three 20-bit counts): closed form (<= 2 violated entries), scans on register columns, general loop (>= 4 columns)
pk = out[:, 7].astype(np.int64)
fast = pk & 0xFFFFF; scan = (pk >> 20) & 0xFFFFF; old = pk >> 40
heavy = np.argsort(-rounds)[:8]
print("round kinds over all runs: closed form %d, register scans %d, general loop %d" % (fast.sum(), scan.sum(), old.sum()))
for k in heavy:
    print("  run %4d: rounds %5d = closed %4d + scans %4d + general %4d ; limit Mcycles %.2f of %.2f" % (k, rounds[k], fast[k], scan[k], old[k], jl[k] / 1e6, tot_all[k] / 1e6))
