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


def _xworker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sqmc_amd import host as H
    # rank r sends (r+1) records to rank 0, (r+2) to rank 1, ...; record = [src, dst, seq, 0]
    counts = [rank + 1 + d for d in range(world)]
    rows = [[rank, d, k, 0] for d in range(world) for k in range(counts[d])]
    send = torch.tensor(rows + [[-1] * 4] * 3, dtype=torch.int64)
    recv = torch.empty((64, 4), dtype=torch.int64)
    nr = H.exchange_records(send, counts, recv)
    q.put((rank, nr, recv[:nr].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_spawn_record_exchange_three_ranks():
    """exchange_records = the all-to-all of mpi_sendnewwalks: rows arrive grouped by source rank,
    in the sender's order."""
    world, port = 3, 29523
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_xworker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps: p.join(60)
    for rank, nr, rows in res:
        expect = [[src, rank, k, 0] for src in range(world) for k in range(src + 1 + rank)]
        assert nr == len(expect) and rows == expect
