"""least squares of a run's joint-limit cycles over its closed-form rounds, scan rounds and iterations (gpurun_out/r05/c4_rounds.npz of c4_rounds_dump.py)"""
import os, numpy as np
d = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "gpurun_out", "r05", "c4_rounds.npz"))
ph = d["phase"]; st = d["status"]; it = d["iters"]
pk = ph[:, 7].astype(np.int64)
closed = pk & 0xFFFFF; scan = (pk >> 20) & 0xFFFFF
ok = st == 0
A = np.stack([closed[ok], scan[ok], it[ok]], 1).astype(float)
x, *_ = np.linalg.lstsq(A, ph[ok, 4], rcond=None)
tot = A @ x
print("runs inside their limits %d: cycles of the joint-limit phase ~ %.0f x closed-form rounds + %.0f x scan rounds + %.0f per iteration" % (ok.sum(), x[0], x[1], x[2]))
print("shares: closed %.0f %%, scans %.0f %%, per iteration %.0f %%; rounds per run: closed %.0f scans %.0f" % (
    100*(A[:, 0]*x[0]).sum()/tot.sum(), 100*(A[:, 1]*x[1]).sum()/tot.sum(), 100*(A[:, 2]*x[2]).sum()/tot.sum(), A[:, 0].mean(), A[:, 1].mean()))
