"""Multi-GPU shard of independent DDP instances (new, no reference counterpart).

The Riccati recursion itself does not shard (sequential in t): what shards are the independent units, the
random-seed instances.  Instance s lives on rank s mod G; every rank runs its instances end to end with zero
communication; the only exchange is the best-cost pick: all-reduce(min) of the cost, then all-reduce(min) of
the masked global index (no MINLOC in RCCL) -- 16 bytes, xGMI-latency bound.

`backend="nccl"` is RCCL on ROCm (one process per GPU); the same code runs on `gloo` for the CPU tests.
"""
import numpy as np

INT64_MAX = np.iinfo(np.int64).max


def instances_of_rank(n_instances, rank, world):
    """global instance indices owned by `rank` (round robin: s -> s mod world)"""
    return list(range(rank, n_instances, world))


def owner_of(instance, world):
    return instance % world


def best_of(local_costs, local_global_indices, device=None):
    """(min cost over all ranks, smallest global index attaining it).  Works without an initialised process
    group (single rank)."""
    import torch
    import torch.distributed as dist
    local_costs = np.asarray(local_costs, dtype=np.float64)
    if local_costs.size:
        j = int(np.argmin(local_costs))
        cost, gidx = float(local_costs[j]), int(local_global_indices[j])
    else:
        cost, gidx = float("inf"), INT64_MAX
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return cost, gidx
    t = torch.tensor([cost], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    best = float(t[0])
    i = torch.tensor([gidx if cost == best else INT64_MAX], dtype=torch.int64, device=device)
    dist.all_reduce(i, op=dist.ReduceOp.MIN)
    return best, int(i[0])
