import ctypes as C
import os

import numpy as np

from ema_amd import clouds as _clouds
from ema_amd import engine as E
from ema_amd import stream as S

SINK = S.SINK
STAT_FIELDS = S.STAT_FIELDS


def default_opts():
    return S.default_opts()


def _deliver(eng, k, bases, off, raw_sink):
    batch, rec, pair_off = eng.align(bases, off)
    n = (len(off) - 1) // 2
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt); keep.append(a); return a
    cand = arr(batch.cand, E.CAND_DTYPE)
    cig = arr(np.append(batch.cigar, 0), np.uint32)
    b = E.BatchOut(n, arr(batch.cand_off, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), cand.ctypes.data_as(C.POINTER(E.Cand)),
                   cig.ctypes.data_as(C.POINTER(C.c_uint32)), len(batch.cigar), 0, arr(batch.status, np.int32).ctypes.data_as(C.POINTER(C.c_int32)),
                   arr(np.zeros(1), np.uint32).ctypes.data_as(C.POINTER(C.c_uint32)))
    r = arr(rec, E.ALN_REC_DTYPE)
    a = E.AlnOut(n, len(rec), arr(pair_off, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), r.ctypes.data_as(C.POINTER(E.AlnRec)))
    rc = raw_sink(None, k, None, C.pointer(b), C.pointer(a)) if raw_sink is not None else 0
    assert rc == 0
    flags = 0
    if os.environ.get("BENCH_STUB_FLAG_RANK") == os.environ.get("RANK", "0"):
        flags = 16      # a read over a capacity on this rank only: every rank must still leave together
    per_read = np.diff(batch.cand_off.astype(np.int64))
    return dict(pairs=n, candidates=int(batch.cand_off[-1]), reads_with_candidates=int((per_read > 0).sum()), records=len(rec),
                unique_records=int(rec["unique"].sum()) if len(rec) else 0, redone_pairs=0, barcode_groups=0, mapq_hist=[0] * 7,
                capacity_flags=flags, rc=0, read_s=0.0, align_s=0.01, append_s=0.001, seed_ms=1.0, extend_ms=1.0, rescue_ms=1.0, final_ms=1.0,
                full_tier_ms=1.0)


def stream_resident(eng, offs, slots_per_set, sink=None, opts=None, raw_sink=None):
    peer = eng.peer() if (opts is not None and opts.n_engines == 2) else None
    sets = [eng, peer] if peer is not None else [eng]
    out = []
    for k, off in enumerate(offs):
        g = sets[k % len(sets)]
        bases, soff = g.slots[(k // len(sets)) % slots_per_set]
        assert (soff == off).all()
        out.append(_deliver(eng, k, bases, soff, raw_sink))
    return out


def stream_batches(eng, batches, sink=None, opts=None, raw_sink=None):
    return [_deliver(eng, k, np.asarray(b), np.asarray(o), raw_sink) for k, (b, o) in enumerate(batches)]
