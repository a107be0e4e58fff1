"""-m gpu: named regressions -- faults and wrong answers the wider, seeded tests found once, pinned here with the smallest case that shows them.

1. Round 5: the update phase read its gradient rows four entries ahead when they live in global memory, and the compiler split the loop's
   constant `q*BLOCK` off the index arithmetic of the band form of A T + B (derivative >= 2) into the immediate offset of FLAT accesses: the base
   fell below the trajectory at the start of LDS, outside the LDS aperture, and the kernel faulted (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION; draw 14
   of tests/test_gpu_random_robots.py; DESIGN.md section 3 "Round 5").  Needs: derivative 2, more than four entries per thread (m n > 1024), the
   gradient rows in global memory."""
import numpy as np
import pytest

import common
import or_cdchomp_amd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("momentum", [0, 1])
def test_second_order_metric_long_trajectory_rows_in_global_memory(oracle, momentum, monkeypatch):
    """WAM, 160 waypoints (158 x 7 = 1106 entries: five per thread), derivative 2, gradient rows forced into global memory: runs to its
    end and matches the oracle (src/libcd/chomp.c:430-683 with the pentadiagonal metric of src/libcd/chomp.c:261-330)"""
    monkeypatch.setenv("ORC_G_LDS", "0")
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    goals = common.wam_goals(6, seed=77)
    kw = dict(n_points=160, lambda_=200.0, obs_factor=100.0, derivative=2, use_momentum=momentum)
    bid = mod.batch_create(model.name, goals, **kw)
    costs, status = mod.batch_iterate(bid, 12)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    mod.close()
    monkeypatch.delenv("ORC_G_LDS")
    _, base, dofvals, adofs = common.wam_state()
    prob = common.tabletop_problem(oracle)
    okw = dict(kw); okw["D"] = okw.pop("derivative")
    rob = oracle.OraRobot(model)
    otraj, ocosts, ost = oracle.batch_run(rob, base, dofvals, adofs, goals, [prob["sdf"]], [prob["pose"]], oracle.default_params(**okw), 12)[:3]
    ok = (status == 0) & (ost == 0)
    assert ok.sum() >= 4, (status, ost)
    err = max(common.rel_l2(traj[k], otraj[k]) for k in np.flatnonzero(ok))
    assert err <= 1e-6, err
    assert np.allclose(costs[ok], ocosts[ok], rtol=1e-6, atol=0)


def test_band_passes_with_any_kernarg_address():
    """round 6: the band passes of `derivative >= 2` runs are called functions that rebuild the (wave-uniform) address of the kernarg
    block from two readfirstlane halves; the first version OR-ed the low half in as an int, which sign-extends: a launch whose kernarg
    block sat at an address with bit 31 set read its batch descriptor from a wild pointer and faulted (half of all launches: the
    address comes from a ring).  Sixteen launches of the same small batch: all finish, all give the same bits."""
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(0)
    model = common.setup_product_wam(mod)
    goals = common.wam_goals(3, seed=8)
    ref = None
    for k in range(16):
        bid = mod.batch_create(model.name, goals, n_points=60, lambda_=100.0, obs_factor=200.0, derivative=2 + (k % 2))
        costs, status = mod.batch_iterate(bid, 2)
        traj = mod.batch_gettraj(bid)
        mod.batch_destroy(bid)
        assert (status == 0).all()
        if k < 2:
            ref = (ref or {}); ref[k % 2] = (traj, costs)
        else:
            assert np.array_equal(traj, ref[k % 2][0]) and np.array_equal(costs, ref[k % 2][1])
    mod.close()
