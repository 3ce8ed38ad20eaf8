"""Host mirror of the bucket loop (include/ema_stream.h): many buckets through reader -> engine -> append stage on one
GPU, in order.  ctypes over the C ABI in libema_engine.so; the sink callback receives numpy copies."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _engine
from . import ingest as _ingest

STAT_FIELDS = ("pairs", "candidates", "reads_with_candidates", "records", "unique_records", "redone_pairs", "barcode_groups")


class StreamOpts(C.Structure):
    _fields_ = [("bc_len", C.c_int), ("is_haplotag", C.c_int), ("max_read_len", C.c_int), ("error_rate", C.c_double),
                ("n_engines", C.c_int), ("read_ahead", C.c_int), ("fastq_input", C.c_int), ("fastq_name_style", C.c_int),
                ("paths2", C.POINTER(C.c_char_p))]


class BucketStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in STAT_FIELDS] + [("mapq_hist", C.c_uint64 * 7), ("capacity_flags", C.c_int32), ("rc", C.c_int32),
                                                          ("read_s", C.c_double), ("align_s", C.c_double), ("append_s", C.c_double)] + \
               [(n, C.c_float) for n in ("seed_ms", "extend_ms", "rescue_ms", "final_ms", "full_tier_ms", "pad_")]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n in STAT_FIELDS}
        d.update(mapq_hist=[int(x) for x in self.mapq_hist], capacity_flags=int(self.capacity_flags), rc=int(self.rc),
                 read_s=float(self.read_s), align_s=float(self.align_s), append_s=float(self.append_s),
                 seed_ms=float(self.seed_ms), extend_ms=float(self.extend_ms), rescue_ms=float(self.rescue_ms),
                 final_ms=float(self.final_ms), full_tier_ms=float(self.full_tier_ms))
        return d


SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(_ingest._Bucket), C.POINTER(_engine.BatchOut), C.POINTER(_engine.AlnOut))


def _lib():
    L = _engine.load_library()
    if not getattr(L, "_stream_bound", False):
        L.ema_stream_opts_default.argtypes = [C.POINTER(StreamOpts)]
        L.ema_stream_buckets.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(StreamOpts), SINK, C.c_void_p, C.POINTER(BucketStats)]
        L.ema_stream_batches.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t,
                                         C.POINTER(StreamOpts), SINK, C.c_void_p, C.POINTER(BucketStats)]
        L.ema_stream_resident.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_int,
                                          C.POINTER(StreamOpts), SINK, C.c_void_p, C.POINTER(BucketStats)]
        L.ema_stream_last_error.restype = C.c_char_p
        L.ema_host_cpu_seconds.argtypes = [C.POINTER(C.c_double), C.c_int]
        L.ema_host_cpu_seconds.restype = None
        L._stream_bound = True
    return L


HOST_STAGES = ("reader", "staging", "fetch", "append", "clouds_em_duplicates", "formatter_and_write")


def host_cpu_seconds(reset: bool = False) -> dict:
    """ema_host_cpu_seconds: CPU seconds per host stage, summed over the stage's threads, since load or the last reset."""
    out = (C.c_double * 6)()
    _lib().ema_host_cpu_seconds(out, 1 if reset else 0)
    return dict(zip(HOST_STAGES, (float(x) for x in out)))


def default_opts() -> StreamOpts:
    o = StreamOpts()
    _lib().ema_stream_opts_default(C.byref(o))
    return o


def _views(pb, pa):
    """numpy copies of one bucket's candidates and records inside the sink"""
    o = pb.contents
    n = o.n_pairs
    cand_off = np.ctypeslib.as_array(o.cand_off, shape=(2 * n + 1,)).copy()
    n_cand = int(cand_off[-1])
    cand = np.frombuffer(C.string_at(o.cand, n_cand * C.sizeof(_engine.Cand)), dtype=_engine.CAND_DTYPE).copy() if n_cand else np.zeros(0, _engine.CAND_DTYPE)
    cigar = np.ctypeslib.as_array(o.cigar, shape=(max(o.n_cigar, 1),)).copy()[:o.n_cigar]
    status = np.ctypeslib.as_array(o.status, shape=(2 * n,)).copy() if n else np.zeros(0, np.int32)
    redone = np.ctypeslib.as_array(o.redone, shape=(max(int(o.n_redone), 1),)).copy()[:int(o.n_redone)]
    batch = _engine.Batch(cand_off, cand, cigar, status, int(o.n_redone), redone)
    a = pa.contents
    rec = np.frombuffer(C.string_at(a.rec, a.n * C.sizeof(_engine.AlnRec)), dtype=_engine.ALN_REC_DTYPE).copy() if a.n else np.zeros(0, _engine.ALN_REC_DTYPE)
    pair_off = np.ctypeslib.as_array(a.pair_off, shape=(a.n_pairs + 1,)).copy()
    return batch, rec, pair_off


def _wrap(sink, want_bucket):
    err = []

    def cb(_user, k, pbk, pb, pa):
        try:
            if sink is None:
                return 0
            batch, rec, pair_off = _views(pb, pa)
            bucket = None
            if want_bucket and pbk:
                bk = pbk.contents
                n = bk.n_pairs
                off = np.ctypeslib.as_array(bk.off, shape=(2 * n + 1,)).copy()
                bases = np.ctypeslib.as_array(C.cast(bk.bases, C.POINTER(C.c_uint8)), shape=(max(int(off[-1]), 1),)).copy()[:int(off[-1])]
                bc = np.ctypeslib.as_array(bk.bc, shape=(max(n, 1),)).copy()[:n]
                bucket = (bases, off, bc)
            sink(int(k), bucket, batch, rec, pair_off)
            return 0
        except BaseException as e:      # noqa: BLE001 -- must not propagate through the C frames
            err.append(e)
            return -100
    return SINK(cb), err


def stream_buckets(eng: "_engine.Engine", paths, sink=None, opts: StreamOpts | None = None):
    """ema_stream_buckets: sink(k, (bases, off, bc), Batch, records, pair_off) per bucket in order.  Returns per-bucket stats."""
    L = _lib()
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    stats = (BucketStats * max(len(paths), 1))()
    cb, err = _wrap(sink, True)
    rc = L.ema_stream_buckets(eng._h, arr, len(paths), C.byref(opts) if opts is not None else None, cb, None, stats)
    if err:
        raise err[0]
    if rc != 0:
        raise RuntimeError(f"ema_stream_buckets failed ({rc}): {L.ema_stream_last_error().decode()}")
    return [stats[k].as_dict() for k in range(len(paths))]


def stream_batches(eng: "_engine.Engine", batches, sink=None, opts: StreamOpts | None = None, raw_sink=None):
    """ema_stream_batches on [(bases u8, off u32), ...] held in host memory.  Returns per-batch stats."""
    L = _lib()
    n = len(batches)
    keep = [(np.ascontiguousarray(b, dtype=np.uint8), np.ascontiguousarray(o, dtype=np.uint32)) for b, o in batches]
    pb = (C.c_void_p * max(n, 1))(*[b.ctypes.data for b, _ in keep])
    po = (C.c_void_p * max(n, 1))(*[o.ctypes.data for _, o in keep])
    pn = (C.c_size_t * max(n, 1))(*[(len(o) - 1) // 2 for _, o in keep])
    stats = (BucketStats * max(n, 1))()
    if raw_sink is not None:
        cb, err = raw_sink, []
    else:
        cb, err = _wrap(sink, False)
    rc = L.ema_stream_batches(eng._h, pb, po, pn, n, C.byref(opts) if opts is not None else None, cb, None, stats)
    if err:
        raise err[0]
    if rc != 0:
        raise RuntimeError(f"ema_stream_batches failed ({rc}): {L.ema_stream_last_error().decode()}")
    return [stats[k].as_dict() for k in range(n)]


def stream_resident(eng: "_engine.Engine", offs, slots_per_set: int, sink=None, opts: StreamOpts | None = None, raw_sink=None):
    """ema_stream_resident: batch k was staged with stage_slot (see the header); offs[k] = its read offsets.
    raw_sink: a ready SINK callback (ctypes pointers, no copies) instead of `sink`."""
    L = _lib()
    n = len(offs)
    keep = [np.ascontiguousarray(o, dtype=np.uint32) for o in offs]
    po = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in keep])
    pn = (C.c_size_t * max(n, 1))(*[(len(o) - 1) // 2 for o in keep])
    stats = (BucketStats * max(n, 1))()
    if raw_sink is not None:
        cb, err = raw_sink, []
    else:
        cb, err = _wrap(sink, False)
    rc = L.ema_stream_resident(eng._h, po, pn, n, slots_per_set, C.byref(opts) if opts is not None else None, cb, None, stats)
    if err:
        raise err[0]
    if rc != 0:
        raise RuntimeError(f"ema_stream_resident failed ({rc}): {L.ema_stream_last_error().decode()}")
    return [stats[k].as_dict() for k in range(n)]


class SamRunOpts(C.Structure):
    pass


def platform_opts(name: str) -> dict:
    """ema_sam_run_opts_platform: what `ema align -p <name>` takes from the reference's platform table (src/techs.c:74-135)."""
    from . import clouds as _clouds
    from . import sam as _sam
    if not SamRunOpts.__dict__.get("_fields_"):
        SamRunOpts._fields_ = [("stream", StreamOpts), ("clouds", _clouds.CloudOpts), ("sam", _sam.SamOpts), ("continue_cloud_ids", C.c_int32)]
    L = _lib()
    L.ema_sam_run_opts_platform.argtypes = [C.c_char_p, C.POINTER(SamRunOpts)]
    L.ema_sam_run_opts_platform.restype = C.c_int
    o = SamRunOpts()
    rc = L.ema_sam_run_opts_platform(name.encode(), C.byref(o))
    if rc != 0:
        raise ValueError(f"unknown platform {name!r}")
    return {"bc_len": o.stream.bc_len, "is_haplotag": bool(o.stream.is_haplotag), "error_rate": o.stream.error_rate,
            "dist_thresh": o.clouds.dist_thresh, "many_clouds": bool(o.clouds.many_clouds), "sam_bc_len": o.sam.bc_len, "sam_is_haplotag": bool(o.sam.is_haplotag),
            "fastq_name_style": o.stream.fastq_name_style, "density_probs": [float(o.clouds.density_probs[i]) for i in range(o.clouds.n_density_probs)]}


def stream_sam(eng: "_engine.Engine", paths, fd: int, rg_id: bytes | None = None, is_haplotag: bool = False, bc_len: int = 16,
               continue_cloud_ids: bool = False, n_engines: int = 0, bx_index: bytes | None = None, density_opt: bool = False,
               fastq_mates: list | None = None, platform: str | None = None, density_seed: int | None = None):
    """ema_stream_sam: bucket files -> SAM text on fd.  Returns (per-bucket stream stats, per-bucket SAM stats).
    platform: `-p <name>` (ema_sam_run_opts_platform; is_haplotag / bc_len are then the platform's).
    density_seed (with density_opt, without continue_cloud_ids): every bucket draws -d's moves from a stream of its own, bucket k's
    seeded with density_seed + k (ema_cloud_opts.seed_private: the bucket as its own `ema align -s` process); None: the process's one
    rand() stream (ema_amd.clouds.reseed), buckets one after the other."""
    from . import clouds as _clouds
    from . import sam as _sam
    if not SamRunOpts.__dict__.get("_fields_"):
        SamRunOpts._fields_ = [("stream", StreamOpts), ("clouds", _clouds.CloudOpts), ("sam", _sam.SamOpts), ("continue_cloud_ids", C.c_int32)]
    L = _lib()
    L.ema_sam_run_opts_default.argtypes = [C.POINTER(SamRunOpts)]
    L.ema_stream_sam.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_size_t, C.POINTER(SamRunOpts), C.c_int, C.POINTER(BucketStats),
                                 C.POINTER(_clouds.SamStats)]
    o = SamRunOpts()
    if platform is not None:
        L.ema_sam_run_opts_platform.argtypes = [C.c_char_p, C.POINTER(SamRunOpts)]
        if L.ema_sam_run_opts_platform(platform.encode(), C.byref(o)) != 0:
            raise ValueError(f"unknown platform {platform!r}")
        o.stream.n_engines = n_engines
    else:
        L.ema_sam_run_opts_default(C.byref(o))
        o.stream.is_haplotag, o.stream.bc_len, o.stream.n_engines = int(is_haplotag), bc_len, n_engines
    o.sam.rg_id = rg_id
    if bx_index is not None:
        o.sam.bx_index = bx_index
    o.continue_cloud_ids = int(continue_cloud_ids)
    o.clouds.density_opt = int(density_opt)
    if density_seed is not None:
        o.clouds.seed_private, o.clouds.seed = 1, int(density_seed) & 0xffffffff
    keep2 = None
    if fastq_mates is not None:      # `ema align -1 [-2]`: paths are FASTQ files; fastq_mates[k] = the mate-2 file of paths[k] or None (interleaved)
        o.stream.fastq_input = 1
        if any(m is not None for m in fastq_mates):
            keep2 = (C.c_char_p * len(paths))(*[m.encode() if m else None for m in fastq_mates])
            o.stream.paths2 = keep2
    arr = (C.c_char_p * max(1, len(paths)))(*[p.encode() for p in paths])
    bst = (BucketStats * max(len(paths), 1))()
    sst = (_clouds.SamStats * max(len(paths), 1))()
    rc = L.ema_stream_sam(eng._h, arr, len(paths), C.byref(o), fd, bst, sst)
    if rc != 0:
        raise RuntimeError(f"ema_stream_sam failed ({rc}): {L.ema_stream_last_error().decode()}")
    return [bst[k].as_dict() for k in range(len(paths))], [sst[k].as_dict() for k in range(len(paths))]
