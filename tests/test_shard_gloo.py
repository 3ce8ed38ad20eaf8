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


def _bucket_record(b):
    """Stand-in for "bucket b aligned and printed on my GPU": an ema_bucket_stats dict and an ema_sam_stats dict (the shapes
    ema_amd.stream.stream_sam returns) whose every field identifies the bucket and the field."""
    st = {n: 1000 * (k + 1) + b for k, n in enumerate(shard.STREAM_COUNTERS)}
    st["mapq_hist"] = [10 * b + k for k in range(shard.MAPQ_BINS)]
    st["capacity_flags"] = b & 1
    st.update(read_s=0.25 * b, align_s=0.5 + b, append_s=0.125)
    st.update(seed_ms=1.5 * b, extend_ms=2.0, rescue_ms=0.001 * b, final_ms=3.25, full_tier_ms=40.0 + b)
    sam = {n: 7000 * (k + 1) + b for k, n in enumerate(shard.SAM_COUNTERS)}
    sam["mapq_hist"] = [100 * b + k for k in range(shard.MAPQ_BINS)]
    sam.update(select_s=0.75 * b, write_s=0.0625)
    return st, sam


def _worker(rank, world, port, n_buckets, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.buckets_of_rank(n_buckets, world, rank)
    # stand-in for "align bucket b on my GPU": statistics that identify the bucket
    local = np.array([shard.bucket_stats(*_bucket_record(b)) for b in mine], dtype=np.int64).reshape(-1, len(shard.STAT_FIELDS))
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
    expect = [shard.bucket_stats(*_bucket_record(b)).tolist() for b in range(n_buckets)]
    for rank, mine, table in results:
        assert mine == list(range(rank, n_buckets, world))
        assert table == expect, f"rank {rank} gathered {table}"
    # bucket by bucket, field by field: rank 0's table holds every field of ema_bucket_stats and of ema_sam_stats
    table = [r for r in results if r[0] == 0][0][2]
    for b in range(n_buckets):
        st, sam = _bucket_record(b)
        d = shard.stats_as_dict(table[b])
        for n in shard.STREAM_COUNTERS:
            assert d[n] == st[n]
        assert d["mapq_hist"] == st["mapq_hist"] and d["capacity_flags"] == st["capacity_flags"]
        assert (d["read_us"], d["align_us"], d["append_us"]) == (250000 * b, 500000 + 1000000 * b, 125000)
        assert (d["seed_us"], d["extend_us"], d["rescue_us"], d["final_us"], d["full_tier_us"]) == (1500 * b, 2000, b, 3250, 40000 + 1000 * b)
        for n in shard.SAM_COUNTERS:
            assert d["sam_" + n] == sam[n]
        assert d["sam_mapq_hist"] == sam["mapq_hist"] and (d["sam_select_us"], d["sam_write_us"]) == (750000 * b, 62500)


def test_the_record_names_every_field_of_both_structs():
    """include/ema_stream.h's ema_bucket_stats and include/ema_clouds.h's ema_sam_stats, as ema_amd's ctypes mirrors declare them."""
    from ema_amd import clouds, stream
    want = set()
    for n, _t in stream.BucketStats._fields_:
        if n in ("rc", "pad_"):
            continue      # the call's return code for the bucket travels as an exception, not as a statistic; padding
        want |= {f"mapq_hist_{k}" for k in range(7)} if n == "mapq_hist" else {n[:-2] + "_us"} if n.endswith("_s") else {n[:-3] + "_us"} if n.endswith("_ms") else {n}
    for n, _t in clouds.SamStats._fields_:
        want |= {f"sam_mapq_hist_{k}" for k in range(7)} if n == "mapq_hist" else {"sam_" + n[:-2] + "_us"} if n.endswith("_s") else {"sam_" + n}
    assert want == set(shard.STAT_FIELDS), want ^ set(shard.STAT_FIELDS)


def test_gather_without_process_group_is_identity():
    local = np.arange(2 * len(shard.STAT_FIELDS), dtype=np.int64).reshape(2, -1)
    assert (shard.gather_stats(local, 2) == local).all()
