"""N>1 path on CPU: two gloo ranks exercise the per-step reduction the multi-GPU bench uses
(do_walk.f90:2778) and the rank-offset seeding rule (do_walk.f90:234)."""
import os
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sqmc_amd import host as H
    out = np.arange(16, dtype=np.float64) * (rank + 1)
    red = H.allreduce_step_sums(out)
    seed = H.rank_seed((1346, 5634, 6635, 4361), rank)
    q.put((rank, red.tolist(), seed))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_sum_reduction():
    world, port = 2, 29517
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps: p.join(60)
    base = np.arange(16, dtype=np.float64)
    for rank, red, seed in res:
        assert red[:7] == (base[:7] * 3).tolist()                    # summed over the two ranks
        assert red[7:] == (base[7:] * (rank + 1)).tolist()           # rank-local pieces untouched
    assert res[0][2] != res[1][2] and res[0][2][:3] == res[1][2][:3]


def test_single_process_reduction_is_identity():
    from sqmc_amd import host as H
    out = np.linspace(0, 1, 16)
    assert np.array_equal(H.allreduce_step_sums(out), out)
