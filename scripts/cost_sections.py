"""Cycles of wavefront 0 per section of the 16-lane cost phase, one run alone on the chip
(library built with -DORC_COST_TIMERS).   ORC_LIB=... python scripts/cost_sections.py"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import common, or_cdchomp_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mod = or_cdchomp_amd.Module(0)
model = common.setup_product_wam(mod)
bid = mod.batch_create(model.name, common.wam_goals(n), n_points=100, lambda_=100.0, obs_factor=500.0)
mod.batch_iterate(bid, 50)
