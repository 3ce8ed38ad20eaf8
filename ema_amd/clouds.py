"""Host mirror of the cloud / EM / duplicate stage (include/ema_clouds.h): the reference's find_clouds_and_align() body
after append_alignments (reference src/align.c:347-608, src/samdict.c) for one bucket.  ctypes over the C ABI in
libema_engine.so; the arrays the result points into are kept alive by the returned object."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as _engine
from . import ingest as _ingest
from . import sam as _sam


class CloudOpts(C.Structure):
    _fields_ = [("dist_thresh", C.c_uint32), ("many_clouds", C.c_int32), ("n_threads", C.c_int32), ("first_cloud_id", C.c_int32),
                ("density_opt", C.c_int32), ("n_density_probs", C.c_int32), ("density_probs", C.c_double * 16), ("emit", C.c_int32), ("seed_private", C.c_int32), ("seed", C.c_uint32), ("pad_", C.c_int32)]


class SamStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("groups", "clouds", "bad_clouds", "lines", "mapped", "unmapped_mates", "proper", "duplicates", "with_xa")] + \
               [("mapq_hist", C.c_uint64 * 7), ("select_s", C.c_double), ("write_s", C.c_double)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_ if n not in ("mapq_hist", "select_s", "write_s")}
        d["mapq_hist"] = [int(x) for x in self.mapq_hist]
        d["select_s"], d["write_s"] = float(self.select_s), float(self.write_s)
        return d


class CloudsOut(C.Structure):
    _fields_ = [("n_lines", C.c_size_t), ("lines", C.POINTER(_sam.SamLine)), ("n_recs", C.c_size_t), ("recs", C.POINTER(_sam.SamRec)),
                ("alts", C.POINTER(_sam.SamAlt)), ("idents", C.c_void_p), ("next_cloud_id", C.c_int32), ("stats", SamStats),
                ("n_descs", C.c_size_t), ("n_xas", C.c_size_t), ("n_sel", C.c_size_t), ("descs", C.c_void_p), ("xas", C.c_void_p), ("sel_at", C.c_void_p),
                ("cigar_lo", C.c_uint64), ("cigar_hi", C.c_uint64)]


def _lib():
    L = _engine.load_library()
    if not getattr(L, "_clouds_bound", False):
        L.ema_cloud_opts_default.argtypes = [C.POINTER(CloudOpts)]
        L.ema_clouds_select.argtypes = [C.POINTER(_ingest._Bucket), C.POINTER(_engine.BatchOut), C.POINTER(_engine.AlnOut), C.POINTER(C.c_char_p),
                                        C.c_int32, C.POINTER(CloudOpts), C.POINTER(C.POINTER(CloudsOut))]
        L.ema_clouds_free.argtypes = [C.POINTER(CloudsOut)]
        L._clouds_bound = True
    return L


def reseed(seed: int) -> None:
    """ema_clouds_reseed: srand(seed) for -d, as the reference does once per process with time()."""
    L = _lib()
    L.ema_clouds_reseed.argtypes = [C.c_uint]
    L.ema_clouds_reseed(seed)


def default_opts() -> CloudOpts:
    o = CloudOpts()
    _lib().ema_cloud_opts_default(C.byref(o))
    return o


class Selection:
    """Result of select(): `.lines` / `.n_lines` go straight to ema_amd.sam.format_lines / write_lines."""

    def __init__(self, ptr, keep):
        self._p, self._keep = ptr, keep
        o = ptr.contents
        self.lines, self.n_lines, self.n_recs = o.lines, int(o.n_lines), int(o.n_recs)
        # the compact form (opts.emit 1 / 2): ema_sam_desc[], ema_sam_xa[], first descriptor of every selected pair, the CIGAR range
        self.descs, self.xas, self.sel_at = o.descs, o.xas, o.sel_at
        self.n_descs, self.n_xas, self.n_sel = int(o.n_descs), int(o.n_xas), int(o.n_sel)
        self.cigar_lo, self.cigar_hi = int(o.cigar_lo), int(o.cigar_hi)
        self.next_cloud_id = int(o.next_cloud_id)
        self.stats = o.stats.as_dict()

    def pairs(self):
        """[(record dict, mate dict | None)] in print order (every second line)."""
        out = []
        for i in range(0, self.n_lines, 2):
            ln = self.lines[i]
            out.append((_rec_dict(ln.rec.contents), _rec_dict(ln.mate.contents) if ln.mate else None))
        return out

    def close(self):
        if self._p:
            _lib().ema_clouds_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _rec_dict(r):
    d = dict(ident=r.ident, chrom_id=int(r.chrom_id), pos=int(r.pos), mate=int(r.mate), rev=int(r.rev), duplicate=int(r.duplicate), gamma=float(r.gamma),
             cloud_id=int(r.cloud_id), cloud_bad=int(r.cloud_bad), mapq=int(r.mapq), score_mapq=int(r.score_mapq), n_alts=int(r.n_alts))
    if r.n_alts:
        a = r.alts[0]
        d["alt"] = (a.chrom, int(a.pos), int(a.edit_dist), int(a.rev), [int(a.cigar[i]) for i in range(a.n_cigar)])
    return d


def c_structs(bucket: "_ingest.Bucket", batch: "_engine.Batch", rec: np.ndarray, pair_off: np.ndarray):
    """ctypes views (ema_bucket, ema_batch_out, ema_aln_out) of numpy-held results, and the arrays to keep alive."""
    keep = []

    def arr(a, dt):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a
    n = bucket.n_pairs
    bk = _ingest._Bucket()
    bk.n_pairs, bk.n_groups = n, len(bucket.group_off) - 1
    bk.group_off = arr(bucket.group_off, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64))
    bk.bc = arr(bucket.bc, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64))
    bk.off = arr(bucket.off, np.uint32).ctypes.data_as(C.POINTER(C.c_uint32))
    bk.bases = C.cast(arr(np.append(bucket.bases, 0), np.uint8).ctypes.data, C.POINTER(C.c_char))
    bk.quals = C.cast(arr(np.append(bucket.quals, 0), np.uint8).ctypes.data, C.POINTER(C.c_char))
    bk.id_off = arr(bucket.id_off, np.uint32).ctypes.data_as(C.POINTER(C.c_uint32))
    bk.ids = C.cast(arr(np.append(bucket.ids, 0), np.uint8).ctypes.data, C.POINTER(C.c_char))
    cand = arr(batch.cand, _engine.CAND_DTYPE)
    cig = arr(np.append(batch.cigar, 0), np.uint32)
    b = _engine.BatchOut(n, arr(batch.cand_off, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), cand.ctypes.data_as(C.POINTER(_engine.Cand)),
                         cig.ctypes.data_as(C.POINTER(C.c_uint32)), len(batch.cigar), 0,
                         arr(batch.status, np.int32).ctypes.data_as(C.POINTER(C.c_int32)), None)
    r = arr(rec, _engine.ALN_REC_DTYPE)
    a = _engine.AlnOut(n, len(rec), arr(pair_off, np.uint64).ctypes.data_as(C.POINTER(C.c_uint64)), r.ctypes.data_as(C.POINTER(_engine.AlnRec)))
    return bk, b, a, keep


def select(bucket, batch, rec, pair_off, contig_names, opts: CloudOpts | None = None) -> Selection:
    """ema_clouds_select on numpy-held results of one bucket (reader, engine, append stage)."""
    L = _lib()
    bk, b, a, keep = c_structs(bucket, batch, rec, pair_off)
    names = (C.c_char_p * max(1, len(contig_names)))(*[n if isinstance(n, bytes) else n.encode() for n in contig_names])
    keep.append(names)
    p = C.POINTER(CloudsOut)()
    rc = L.ema_clouds_select(C.byref(bk), C.byref(b), C.byref(a), names, len(contig_names), C.byref(opts) if opts is not None else None, C.byref(p))
    if rc != 0:
        if p:
            L.ema_clouds_free(p)
        raise RuntimeError(f"ema_clouds_select failed ({rc})")
    keep += [bk, b, a]
    sel = Selection(p, keep)
    sel.bk, sel.b = bk, b      # (the bucket and the batch as the C structs: the device formatter's other inputs)
    return sel
