"""one launch of the WAM holding the four-sphere box (config 2's goals), for counter passes:
python scripts/run_held4.py [n_runs=1024] [n_iter=100]   (WGS_PER_CU=4: the kernels built for four workgroups per CU)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import common, or_cdchomp_amd
n_runs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mod = or_cdchomp_amd.Module(0)
mod.set_workgroups_per_cu(int(os.environ.get('WGS_PER_CU', '0')))
model, hand, pose = common.setup_product_wam_held4(mod)
bid = mod.batch_create(model.name, common.wam_goals(n_runs), **common.CONFIG2_KW)
mod.kernel_time(reset=True)
costs, status = mod.batch_iterate(bid, n_iter)
ms, n = mod.kernel_time()
print("held4: %d runs x %d iterations: kernel %.2f ms, %.3g it/s" % (n_runs, n_iter, ms, int(mod.batch_iterations_done(bid).sum()) / (ms * 1e-3)))
