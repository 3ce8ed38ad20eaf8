"""The N>1 path on CPU: two processes (gloo, 127.0.0.1) deal buckets round-robin and gather their statistics.
The data path itself has no collective (buckets are independent), so this covers everything that is
distributed about the engine."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ema_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_buckets, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.buckets_of_rank(n_buckets, world, rank)
    # stand-in for "align bucket b on my GPU": statistics that identify the bucket
    local = np.array([[1000 + b, 3 * b, 2 * b, b, 0, 1, 7] for b in mine], dtype=np.int64).reshape(-1, len(shard.STAT_FIELDS))
    table = shard.gather_stats(local, n_buckets)
    q.put((rank, mine, table.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_deal_covers_every_bucket_once():
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 8, 500):
            seen = sorted(b for r in range(world) for b in shard.buckets_of_rank(n, world, r))
            assert seen == list(range(n))


def test_gather_stats_world_size_2_gloo():
    world, n_buckets = 2, 7            # uneven: rank 0 gets 4 buckets, rank 1 gets 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_buckets, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [[1000 + b, 3 * b, 2 * b, b, 0, 1, 7] for b in range(n_buckets)]
    for rank, mine, table in results:
        assert mine == list(range(rank, n_buckets, world))
        assert table == expect, f"rank {rank} gathered {table}"


def test_gather_without_process_group_is_identity():
    local = np.arange(2 * len(shard.STAT_FIELDS), dtype=np.int64).reshape(2, -1)
    assert (shard.gather_stats(local, 2) == local).all()
