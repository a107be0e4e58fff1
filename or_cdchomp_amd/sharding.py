"""Sharding of a batch of independent runs over the GPUs of one node.

Runs share only read-only data (robot, fields, metric), so the batch is cut into
contiguous blocks, one per rank, with no data-path collective; the
only exchange is a host-side gather of trajectories / costs / status on rank 0 and a
max-reduce of the elapsed time.  One process per GPU, launched by torch.distributed.run.
"""
import numpy as np


def shard_bounds(n_total, rank, world):
    """contiguous block [lo, hi) of rank; the first n_total % world ranks get one extra run"""
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def shard(array, rank, world):
    lo, hi = shard_bounds(len(array), rank, world)
    return array[lo:hi]


def host_group(dist):
    """a gloo group for host-side exchanges next to the (RCCL) default group"""
    if dist is None or not dist.is_initialized():
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    return dist.new_group(backend="gloo")


def gather_host(local, dist, group=None, dst=0):
    """gather a dict of numpy arrays (run-major, first axis = runs) on rank dst; other ranks get None"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: np.asarray(v) for k, v in local.items()}
    world = dist.get_world_size()
    rank = dist.get_rank()
    parts = [None] * world if rank == dst else None
    dist.gather_object({k: np.asarray(v) for k, v in local.items()}, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}


def max_over_ranks(value, dist, group=None):
    """the job's elapsed time is the slowest rank's"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    if dist.get_backend(group) == "nccl":
        t = torch.tensor([value], dtype=torch.float64, device="cuda")
    else:
        t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
