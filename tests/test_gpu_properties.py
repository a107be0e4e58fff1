"""-m gpu: size-independent properties of the HIP path at BASELINE.json's full sizes
(configs[1]: WAM, 100 waypoints, 100 iterations, batch 1024; configs[2]'s sharding rule).

The oracle takes ~0.1 s per run, so at these sizes the checks are properties the domain offers:
runs are independent (a run's result does not depend on what shares its batch, nor on its position,
nor on the stream its launch is issued on), iterating in two calls is iterating in one, shards of a
batch are the batch, and the reported costs are the costs of the returned trajectory."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

KW = dict(n_points=100, lambda_=100.0, obs_factor=500.0)


@pytest.fixture(scope="module")
def full(gpu_module):
    mod = gpu_module
    try:
        model = common.setup_product_wam(mod)
    except RuntimeError:                      # the session's module already holds the scene
        model, _, _, _ = common.wam_state()
    goals = common.wam_goals(1024, seed=20250101)
    bid = mod.batch_create(model.name, goals, **KW)
    costs, status = mod.batch_iterate(bid, 100)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    return dict(mod=mod, model=model, goals=goals, costs=costs, status=status, traj=traj)


def test_runs_are_independent_of_batch_and_position(full):
    """a permuted subset, launched on its own, reproduces the full batch bit for bit"""
    mod, model, goals = full["mod"], full["model"], full["goals"]
    pick = np.random.default_rng(1).permutation(1024)[:200]
    bid = mod.batch_create(model.name, goals[pick], **KW)
    costs, status = mod.batch_iterate(bid, 100)
    traj = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    assert np.array_equal(status, full["status"][pick])
    assert np.array_equal(traj, full["traj"][pick])
    assert np.array_equal(costs, full["costs"][pick])


def test_contiguous_shards_are_the_batch(full):
    """contiguous blocks of runs per GPU; two shards computed separately equal the batch"""
    mod, model, goals = full["mod"], full["model"], full["goals"]
    from or_cdchomp_amd import sharding
    parts = []
    for r in range(2):
        lo, hi = sharding.shard_bounds(1024, r, 2)
        bid = mod.batch_create(model.name, goals[lo:hi], **KW)
        mod.batch_iterate(bid, 100)
        parts.append(mod.batch_gettraj(bid))
        mod.batch_destroy(bid)
    assert np.array_equal(np.concatenate(parts), full["traj"])


@pytest.mark.parametrize("momentum", [0, 1])
def test_two_calls_are_one(full, momentum):
    """iterate(37) then iterate(63) == iterate(100): the cost-only pass that ends a call leaves the
    state alone, momentum and the leapfrog flag carry over (src/orcdchomp_mod.cpp:2752-2830)"""
    mod, model, goals = full["mod"], full["model"], full["goals"][:512]
    kw = dict(KW, use_momentum=momentum)
    a = mod.batch_create(model.name, goals, **kw)
    ca, sa = mod.batch_iterate(a, 100)
    ta = mod.batch_gettraj(a)
    b = mod.batch_create(model.name, goals, **kw)
    _, sb1 = mod.batch_iterate(b, 37)
    cb, sb2 = mod.batch_iterate(b, 63)
    tb = mod.batch_gettraj(b)
    mod.batch_destroy(a); mod.batch_destroy(b)
    # the status is that of the call (a run that left its limits in the first call iterates again in
    # the second, as in the reference): a run fails in one of the two calls iff it fails in the one
    sb = np.minimum(sb1, sb2)
    ok = (sa == 0) & (sb == 0)
    assert np.array_equal(sa, sb)
    assert np.array_equal(ta[ok], tb[ok])
    assert np.array_equal(ca[ok], cb[ok])


def test_streams_do_not_change_results(full):
    """launches issued on a pool of streams give the results of serial launches"""
    model, goals = full["model"], full["goals"]
    import or_cdchomp_amd
    mod = or_cdchomp_amd.Module(0)            # the pool can only change while the module has no batch
    common.setup_product_wam(mod)
    mod.set_num_streams(3)
    try:
        bids = [mod.batch_create(model.name, goals[k*256:(k+1)*256], **KW) for k in range(4)]
        for bid in bids:
            mod.batch_iterate_async(bid, 100)
        for bid in bids:
            mod.batch_sync(bid)
        traj = np.concatenate([mod.batch_gettraj(bid) for bid in bids])
        for bid in bids:
            mod.batch_destroy(bid)
        # the pool is fixed while batches hold its streams
        bid = mod.batch_create(model.name, goals[:4], **KW)
        with pytest.raises(RuntimeError, match="destroy the existing batches"):
            mod.set_num_streams(2)
        mod.batch_destroy(bid)
    finally:
        mod.set_num_streams(0)
    assert np.array_equal(traj, full["traj"])


def test_costs_are_the_costs_of_the_returned_trajectory(full):
    """a further call with n_iter = 0 (the reference's cost-only pass) reproduces the costs"""
    mod, model, goals = full["mod"], full["model"], full["goals"][:256]
    bid = mod.batch_create(model.name, goals, **KW)
    c1, s1 = mod.batch_iterate(bid, 100)
    t1 = mod.batch_gettraj(bid)
    c2, s2 = mod.batch_iterate(bid, 0)
    t2 = mod.batch_gettraj(bid)
    mod.batch_destroy(bid)
    ok = s1 == 0
    assert np.array_equal(t1, t2)
    assert np.array_equal(c1[ok], c2[ok])
    assert np.allclose(c1[ok, 0], c1[ok, 1] + c1[ok, 2], rtol=1e-15, atol=0)
    # end points never move, interior points stay inside the joint limits of successful runs
    _, _, dofvals, _ = common.wam_state()
    assert np.array_equal(t1[:, 0, :], np.tile(dofvals[:7], (256, 1)))
    # the last row is the reference's s + (g - s)*(N-1)/(N-1), not g itself (src/orcdchomp_mod.cpp:2417-2464)
    assert np.allclose(t1[:, -1, :], goals, rtol=4e-16, atol=1e-15)
    lo = np.asarray(model.limit_lower[:7]); hi = np.asarray(model.limit_upper[:7])
    inside = (t1[ok] >= lo - 1e-12) & (t1[ok] <= hi + 1e-12)
    assert inside.all()
