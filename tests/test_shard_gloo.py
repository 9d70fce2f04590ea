"""The N > 1 path on CPU: world_size-2 gloo process group (the GPU box runs the same code over nccl = RCCL)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_instances, ret):
    import torch.distributed as dist
    from ddp_pinocchio_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.instances_of_rank(n_instances, rank, world)
    # synthetic per-instance costs, a function of the global index only; two instances tie for the minimum
    costs_all = np.array([(7 * g) % 11 + 0.25 for g in range(n_instances)])
    costs_all[5] = costs_all[9] = -3.0
    best, idx = shard.best_of(costs_all[mine], mine)
    ret[rank] = (mine, best, idx)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_and_best_pick_world2():
    world, n = 2, 13
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
        ret = dict(ret)
    owned = sorted(ret[0][0] + ret[1][0])
    assert owned == list(range(n))                               # every instance exactly once
    assert all(g % world == r for r in range(world) for g in ret[r][0])
    for r in range(world):
        assert ret[r][1] == -3.0 and ret[r][2] == 5              # min cost, smallest index among the ties


def test_single_rank_needs_no_process_group():
    from ddp_pinocchio_amd import shard
    assert shard.best_of([3.0, 1.0, 2.0], [10, 11, 12]) == (1.0, 11)
    assert shard.instances_of_rank(5, 0, 1) == [0, 1, 2, 3, 4]
    assert shard.owner_of(7, 4) == 3


def _worker_pick(rank, world, port, n_instances, ret):
    import torch.distributed as dist
    from ddp_pinocchio_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.instances_of_rank(n_instances, rank, world)
    costs_all = np.array([(5 * g) % 13 + 0.5 for g in range(n_instances)])
    costs_all[4] = costs_all[7] = -1.5                       # a tie across the two ranks: the smaller global index wins
    best, idx = shard.pick(costs_all[mine], rank, world)
    # the same answer as round 1's two all-reduces
    best2, idx2 = shard.best_of(costs_all[mine], mine)
    # the winner's "trajectory" to local row 0 of every rank: rows are labelled by their global index
    X = np.stack([np.full(6, float(g)) for g in mine])
    U = np.stack([np.full(3, 100.0 + g) for g in mine])
    root, src = shard.broadcast_winner([X, U], idx, rank, world, dst_local=0)
    ret[rank] = (best, idx, best2, idx2, root, src, X[0].tolist(), U[0].tolist(), X[1].tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_one_collective_pick_and_winner_broadcast_world2():
    """ddp_hip_shard_pick / ddp_hip_shard_broadcast's ownership and root logic (instance s -> rank s mod G, local position
    s div G) on the host mirror over gloo: the GPU box runs the library's RCCL path with the same rule"""
    world, n = 2, 11
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_pick, args=(world, _free_port(), n, ret), nprocs=world, join=True)
        ret = dict(ret)
    for r in range(world):
        best, idx, best2, idx2, root, src, x0, u0, x1 = ret[r]
        assert (best, idx) == (-1.5, 4) and (best2, idx2) == (-1.5, 4)
        assert root == 0 and src == 2                            # global 4 lives on rank 0, third local instance
        assert x0 == [4.0] * 6 and u0 == [104.0] * 3             # every rank now holds the winner in its row 0
        assert x1 == [float(r + world)] * 6                      # the other rows are untouched
