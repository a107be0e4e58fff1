"""not gpu: the N>1 path (contiguous sharding of runs, host-side gather, max-over-ranks timing)
with two gloo processes on the CPU.  The per-rank 'compute' is the oracle so that the gathered
result can be compared with a single-process run of the whole batch."""
import os
import socket

import numpy as np
import pytest

import common


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, goals, out_path):
    import torch.distributed as dist
    from oracle import oracle_py as O
    from or_cdchomp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = sharding.host_group(dist)
    prob = common.tabletop_problem(O)
    model, base, dofvals, adofs = common.wam_state()
    mine = sharding.shard(goals, rank, world)
    p = O.default_params(n_points=20, lambda_=100.0, obs_factor=500.0)
    traj, costs, status, _ = O.batch_run(O.OraRobot(model), base, dofvals, adofs, mine, [prob["sdf"]],
                                         [prob["pose"]], p, 5, max_threads=1)
    elapsed = sharding.max_over_ranks(1.0 + rank, dist, grp)
    whole = sharding.gather_host({"traj": traj, "costs": costs, "status": status}, dist, grp)
    if rank == 0:
        np.savez(out_path, elapsed=elapsed, **whole)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(oracle, tmp_path):
    import torch.multiprocessing as mp
    from or_cdchomp_amd import sharding
    assert [sharding.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [sharding.shard_bounds(65536, r, 8) for r in range(8)][3] == (24576, 32768)     # config 3 blocks
    goals = common.wam_goals(7, seed=5)                # uneven split: 4 + 3
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), goals, out), nprocs=2, join=True)
    got = np.load(out)
    assert float(got["elapsed"]) == 2.0                # max over ranks
    prob = common.tabletop_problem(oracle)
    model, base, dofvals, adofs = common.wam_state()
    p = oracle.default_params(n_points=20, lambda_=100.0, obs_factor=500.0)
    traj, costs, status, _ = oracle.batch_run(oracle.OraRobot(model), base, dofvals, adofs, goals, [prob["sdf"]],
                                              [prob["pose"]], p, 5, max_threads=1)
    assert np.array_equal(got["traj"], traj) and np.array_equal(got["costs"], costs)
    assert np.array_equal(got["status"], status)
