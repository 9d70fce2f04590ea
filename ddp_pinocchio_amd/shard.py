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


def local_position(instance, world):
    """position of global instance `instance` inside its owner's context (instances_of_rank is round robin)"""
    return instance // world


def pick(local_costs, rank, world, device=None):
    """The pick as ONE collective -- the host mirror of ddp_hip_shard_pick (csrc/comm.cpp): local argmin (smallest local
    index among ties), one all-gather of the (cost, global index) pairs, argmin of the G pairs (smallest global index among
    ties).  Local instance j of rank r is global instance r + j * world."""
    import torch
    import torch.distributed as dist
    local_costs = np.asarray(local_costs, dtype=np.float64)
    if local_costs.size:
        j = int(np.argmin(local_costs))
        pair = (float(local_costs[j]), rank + j * world)
    else:
        pair = (float("inf"), INT64_MAX)
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return pair
    # the index travels as an int64 bit pattern inside a float64 (no arithmetic touches it), as on the device
    mine = torch.tensor([pair[0], np.array([pair[1]], dtype=np.int64).view(np.float64)[0]], dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    pairs = [(float(t[0]), int(t[1:2].cpu().numpy().view(np.int64)[0])) for t in out]
    return min(pairs)                                      # lexicographic: cost, then global index


def broadcast_winner(arrays, best_global_index, rank, world, dst_local=0, device=None):
    """The host mirror of ddp_hip_shard_broadcast: `arrays` are this rank's [batch][size] numpy sequences (X, U, FB_*); the
    owner of the winner (rank = index mod G) sends its row index div G, every rank stores it in row dst_local."""
    import torch
    import torch.distributed as dist
    root, src = owner_of(best_global_index, world), local_position(best_global_index, world)
    for a in arrays:
        row = torch.from_numpy(np.ascontiguousarray(a[src] if rank == root else a[dst_local]).copy())
        if device is not None:
            row = row.to(device)
        if world > 1:
            dist.broadcast(row, src=root)
        a[dst_local] = row.cpu().numpy()
    return root, src
