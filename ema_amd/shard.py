"""Bucket sharding across the GPUs of one node and the gather of per-bucket statistics.

EMA's barcode buckets are independent units (reference README.md:127-130 runs one process per bucket), so the
multi-GPU layout is: one process per GPU, bucket b -> rank b mod G, every rank holding a full replica of the
index, and NO collective on the data path.  The only exchange is the all-gather of a small statistics vector
per bucket at the end (RCCL over xGMI on the GPU box: backend "nccl"; the CPU tests use gloo).
"""
from __future__ import annotations

import numpy as np

# per-bucket record = the counters of include/ema_stream.h's ema_bucket_stats (SURVEY 8e's BucketStats), in this order
STAT_FIELDS = ("pairs", "candidates", "reads_with_candidates", "records", "unique_records", "redone_pairs", "barcode_groups")


def buckets_of_rank(n_buckets: int, world: int, rank: int):
    """Round-robin deal: bucket b belongs to rank b mod world."""
    return list(range(rank, n_buckets, world))


def bucket_stats(st: dict) -> np.ndarray:
    """Statistics vector of one streamed bucket (ema_amd.stream's per-bucket dict)."""
    return np.array([st[f] for f in STAT_FIELDS], dtype=np.int64)


def gather_stats(local: np.ndarray, n_buckets: int, device=None):
    """All-gathers the [n_local_buckets, F] statistics of every rank into one [n_buckets, F] table indexed by
    bucket id.  Must be called by every rank of the default process group (or alone, without a group)."""
    import torch
    import torch.distributed as dist
    local = np.asarray(local, dtype=np.int64).reshape(-1, len(STAT_FIELDS))
    if not (dist.is_available() and dist.is_initialized()):
        assert local.shape[0] == n_buckets
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per_rank = (n_buckets + world - 1) // world          # pad so that every rank contributes the same shape
    buf = torch.full((per_rank, len(STAT_FIELDS)), -1, dtype=torch.int64)
    buf[:local.shape[0]] = torch.from_numpy(local)
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    table = np.full((n_buckets, len(STAT_FIELDS)), -1, dtype=np.int64)
    for r, part in enumerate(parts):
        part = part.cpu().numpy()
        for k, b in enumerate(buckets_of_rank(n_buckets, world, r)):
            table[b] = part[k]
    assert (table >= 0).all() or rank != 0 or n_buckets == 0 or (table[:, 0] >= 0).all()
    return table
