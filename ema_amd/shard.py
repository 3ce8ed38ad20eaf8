"""Bucket sharding across the GPUs of one node and the gather of per-bucket statistics.

EMA's barcode buckets are independent units (reference README.md:127-130 runs one process per bucket), so the
multi-GPU layout is: one process per GPU, bucket b -> rank b mod G, every rank holding a full replica of the
index, and NO collective on the data path.  The only exchange is the all-gather of a small statistics vector
per bucket at the end (RCCL over xGMI on the GPU box: backend "nccl"; the CPU tests use gloo).
"""
from __future__ import annotations

import numpy as np

# The per-bucket record the ranks gather (SURVEY 8e's BucketStats: "pairs, records, mapped, proper-paired, duplicates, MAPQ
# histogram, engine seconds") = every field of include/ema_stream.h's ema_bucket_stats followed by every field of
# include/ema_clouds.h's ema_sam_stats, as int64: times in microseconds (the structs carry seconds as double and kernel
# milliseconds as float); a bucket that never went through the cloud stage / formatter has zeros in the SAM part.
STREAM_COUNTERS = ("pairs", "candidates", "reads_with_candidates", "records", "unique_records", "redone_pairs", "barcode_groups")
MAPQ_BINS = 7      # 0, 1-9, 10-19, 20-29, 30-39, 40-59, 60
STREAM_TIMES = ("read_s", "align_s", "append_s")                                       # -> *_us
ENGINE_MS = ("seed_ms", "extend_ms", "rescue_ms", "final_ms", "full_tier_ms")          # -> *_us
SAM_COUNTERS = ("groups", "clouds", "bad_clouds", "lines", "mapped", "unmapped_mates", "proper", "duplicates", "with_xa")
SAM_TIMES = ("select_s", "write_s")                                                     # -> *_us
STAT_FIELDS = (STREAM_COUNTERS + tuple(f"mapq_hist_{k}" for k in range(MAPQ_BINS)) + ("capacity_flags",)
               + tuple(n[:-2] + "_us" for n in STREAM_TIMES) + tuple(n[:-3] + "_us" for n in ENGINE_MS)
               + tuple("sam_" + n for n in SAM_COUNTERS) + tuple(f"sam_mapq_hist_{k}" for k in range(MAPQ_BINS))
               + tuple("sam_" + n[:-2] + "_us" for n in SAM_TIMES))


def buckets_of_rank(n_buckets: int, world: int, rank: int):
    """Round-robin deal: bucket b belongs to rank b mod world."""
    return list(range(rank, n_buckets, world))


def bucket_stats(st: dict, sam: dict | None = None) -> np.ndarray:
    """The record of one streamed bucket: st = ema_amd.stream's per-bucket dict (ema_bucket_stats), sam = the same bucket's
    ema_sam_stats dict when it went on to SAM text (ema_stream_sam), else None."""
    v = [int(st[f]) for f in STREAM_COUNTERS]
    v += [int(x) for x in st.get("mapq_hist", [0] * MAPQ_BINS)]
    v.append(int(st.get("capacity_flags", 0)))
    v += [int(round(float(st.get(n, 0.0)) * 1e6)) for n in STREAM_TIMES]
    v += [int(round(float(st.get(n, 0.0)) * 1e3)) for n in ENGINE_MS]
    sam = sam or {}
    v += [int(sam.get(n, 0)) for n in SAM_COUNTERS]
    v += [int(x) for x in sam.get("mapq_hist", [0] * MAPQ_BINS)]
    v += [int(round(float(sam.get(n, 0.0)) * 1e6)) for n in SAM_TIMES]
    assert len(v) == len(STAT_FIELDS)
    return np.array(v, dtype=np.int64)


def stats_as_dict(row) -> dict:
    """One row of the gathered table back as names -> values (histograms as lists)."""
    d = dict(zip(STAT_FIELDS, (int(x) for x in row)))
    d["mapq_hist"] = [d.pop(f"mapq_hist_{k}") for k in range(MAPQ_BINS)]
    d["sam_mapq_hist"] = [d.pop(f"sam_mapq_hist_{k}") for k in range(MAPQ_BINS)]
    return d


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
