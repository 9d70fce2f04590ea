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
