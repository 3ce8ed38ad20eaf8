"""ctypes binding of oracle/libemaoracle.so -- test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Opt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5",
                                       "pen_clip3", "w", "zdrop")] + [("max_mem_intv", C.c_uint64)] + \
               [(n, C.c_int) for n in ("T", "flag", "min_seed_len", "min_chain_weight", "max_chain_extend")] + \
               [("split_factor", C.c_float)] + \
               [(n, C.c_int) for n in ("split_width", "max_occ", "max_chain_gap", "n_threads", "chunk_size")] + \
               [(n, C.c_float) for n in ("mask_level", "drop_ratio", "XA_drop_ratio", "mask_level_redun",
                                         "mapQ_coef_len")] + \
               [(n, C.c_int) for n in ("mapQ_coef_fac", "max_ins", "max_matesw", "max_XA_hits", "max_XA_hits_alt")] + \
               [("mat", C.c_int8 * 25)]


class Intv(C.Structure):
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


class IntvV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(Intv))]


class Reg(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64)] + \
               [(n, C.c_int) for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
                                       "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] + \
               [("frac_rep", C.c_float), ("hash", C.c_uint64)]


class RegV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(Reg))]


class Cand(C.Structure):
    _fields_ = [("reg", Reg), ("pos", C.c_int64), ("is_rev", C.c_int), ("NM", C.c_int), ("n_cigar", C.c_int),
                ("cigar_off", C.c_uint32), ("aln_score", C.c_int), ("aln_sub", C.c_int)]


class PairOut(C.Structure):
    _fields_ = [("n1", C.c_size_t), ("n2", C.c_size_t), ("c", C.POINTER(Cand)), ("n_pool", C.c_size_t),
                ("pool", C.POINTER(C.c_uint32))]


class Kswr(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_ext", "n_lf", "n_occ", "w_ref", "n_regs", "n_cigar", "l_read",
                                          "cells_ext", "cells_local", "cells_global", "rows_ext", "rows_local",
                                          "rows_global", "n_ext_calls", "n_local_calls", "n_global_calls")]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "libemaoracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.orc_idx_load.restype = C.c_void_p
        L.orc_idx_load.argtypes = [C.c_char_p]
        L.orc_idx_destroy.argtypes = [C.c_void_p]
        L.orc_opt_init.argtypes = [C.POINTER(Opt)]
        L.orc_collect_intv.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p, C.POINTER(IntvV)]
        L.orc_sa.restype = C.c_uint64
        L.orc_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_occ.restype = C.c_uint64
        L.orc_occ.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.orc_align_pair.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_char_p, C.c_int, C.c_char_p, C.c_int,
                                     C.POINTER(PairOut)]
        L.orc_pair_out_free.argtypes = [C.POINTER(PairOut)]
        L.orc_ksw_extend2.restype = C.c_int
        L.orc_ksw_extend2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int8)] + \
            [C.c_int] * 8 + [C.POINTER(C.c_int)] * 5
        L.orc_ksw_global2.restype = C.c_int
        L.orc_ksw_global2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int8)] + \
            [C.c_int] * 5 + [C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint32))]
        L.orc_ksw_align2.restype = Kswr
        L.orc_ksw_align2.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int8)] + \
            [C.c_int] * 5
        L.orc_align1_core.restype = RegV
        L.orc_align1_core.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p]
        L.orc_stats_get.argtypes = [C.POINTER(Stats)]
        L.orc_bench_pairs.restype = C.c_double
        L.orc_bench_pairs.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                      C.POINTER(C.c_uint64)]
        L.orc_introsort_u64.argtypes = [C.c_size_t, C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def default_opt():
    o = Opt()
    lib().orc_opt_init(C.byref(o))
    return o


NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    NT4[_c] = _i
    NT4[_c + 32] = _i


class Index:
    def __init__(self, prefix):
        self.h = lib().orc_idx_load(prefix.encode())
        if not self.h:
            raise RuntimeError(f"oracle could not load index {prefix}")

    def close(self):
        if self.h:
            lib().orc_idx_destroy(self.h)
            self.h = None

    def sa(self, k):
        return lib().orc_sa(self.h, k)


def collect_intv(idx: Index, opt, read_ascii: bytes):
    """SMEM/seed intervals of one read as a list of (start, end, x0, x1, x2)."""
    q = NT4[np.frombuffer(read_ascii, dtype=np.uint8)].tobytes()
    v = IntvV(0, 0, None)
    lib().orc_collect_intv(C.byref(opt), idx.h, len(q), q, C.byref(v))
    out = [(v.a[i].info >> 32, v.a[i].info & 0xffffffff, v.a[i].x[0], v.a[i].x[1], v.a[i].x[2]) for i in range(v.n)]
    C.CDLL(None).free(v.a)
    return out


REG_FIELDS = ("rb", "re", "qb", "qe", "rid", "score", "truesc", "sub", "csub", "sub_n", "w", "seedcov", "secondary",
              "seedlen0", "n_comp", "is_alt", "frac_rep")


def align_pair(idx: Index, opt, r1: bytes, r2: bytes):
    """Returns ([cand dicts of mate1], [cand dicts of mate2]); each has the region fields, pos, is_rev, NM, cigar."""
    out = PairOut()
    lib().orc_align_pair(C.byref(opt), idx.h, r1, len(r1), r2, len(r2), C.byref(out))
    res = ([], [])
    for k in range(out.n1 + out.n2):
        c = out.c[k]
        d = {f: getattr(c.reg, f) for f in REG_FIELDS}
        d.update(pos=c.pos, is_rev=c.is_rev, NM=c.NM,
                 cigar=[out.pool[c.cigar_off + j] for j in range(c.n_cigar)])
        res[0 if k < out.n1 else 1].append(d)
    lib().orc_pair_out_free(C.byref(out))
    return res


def append_alignments(idx: Index, opt, r1: bytes, r2: bytes, error_rate=0.001):
    """The records the reference's append_alignments would emit for this pair: list of dicts (mate, index of the candidate
    within its mate's list, clip, clip_edit_dist, mapq, score_mapq, unique, score)."""
    out = PairOut()
    L = lib()
    L.orc_align_pair(C.byref(opt), idx.h, r1, len(r1), r2, len(r2), C.byref(out))
    n = out.n1 + out.n2
    IA = C.c_int * max(n, 1)
    which, clip, dist, mapq, smq, uniq = IA(), IA(), IA(), IA(), IA(), IA()
    score = (C.c_double * max(n, 1))()
    L.orc_append_alignments.restype = C.c_int
    L.orc_append_alignments.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 7
    k = L.orc_append_alignments(C.byref(opt), C.byref(out), len(r1), len(r2), error_rate, which, clip, dist, mapq, smq, uniq, score)
    res = []
    for i in range(k):
        m = 0 if which[i] < out.n1 else 1
        res.append(dict(mate=m, cand=which[i] - (out.n1 if m else 0), clip=clip[i], clip_edit_dist=dist[i], mapq=mapq[i],
                        score_mapq=smq[i], unique=uniq[i], score=score[i]))
    L.orc_pair_out_free(C.byref(out))
    return res


def align1(idx: Index, opt, read_ascii: bytes):
    """Regions of one read after mem_align1_core (before mate rescue): list of dicts."""
    q = C.create_string_buffer(NT4[np.frombuffer(read_ascii, dtype=np.uint8)].tobytes(), len(read_ascii))
    v = lib().orc_align1_core(C.byref(opt), idx.h, len(read_ascii), q)
    out = [{f: getattr(v.a[i], f) for f in REG_FIELDS} for i in range(v.n)]
    C.CDLL(None).free(v.a)
    return out


def stats_reset():
    lib().orc_stats_reset()


def stats_get():
    s = Stats()
    lib().orc_stats_get(C.byref(s))
    return {n: int(getattr(s, n)) for n, _ in Stats._fields_}


def bench_pairs(idx: Index, opt, bases: np.ndarray, off: np.ndarray, n_threads: int):
    """Times the oracle on a batch (CPU baseline leg of bench.py).  Returns (seconds, candidates)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n_cand = C.c_uint64()
    secs = lib().orc_bench_pairs(C.byref(opt), idx.h, bases.ctypes.data, off.ctypes.data, (len(off) - 1) // 2, n_threads,
                                 C.byref(n_cand))
    return secs, n_cand.value


def digest_pairs(idx: Index, opt, bases: np.ndarray, off: np.ndarray, n_threads: int):
    """Per-read digests (uint64[2 * n_pairs]) of the oracle's candidate lists for a batch (oracle/pair.c, orc_digest_pairs), and the
    seconds it took.  cand_digest() below computes the same digest from candidate arrays."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint32)
    n = (len(off) - 1) // 2
    out = np.zeros(2 * n, dtype=np.uint64)
    L = lib()
    L.orc_digest_pairs.restype = C.c_double
    L.orc_digest_pairs.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    secs = L.orc_digest_pairs(C.byref(opt), idx.h, bases.ctypes.data, off.ctypes.data, n, n_threads, out.ctypes.data)
    return out, secs


_DG = [np.uint64(x) for x in (0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x85EBCA77C2B2AE63, 0x27D4EB2F165667C5,
                              0xD6E8FEB86659FD93, 0xFF51AFD7ED558CCD, 0xC4CEB9FE1A85EC53, 0x2545F4914F6CDD1D, 0x94D049BB133111EB,
                              0xBF58476D1CE4E5B9)]


def cand_digest(cand: np.ndarray, cigar: np.ndarray, read_off: np.ndarray):
    """orc_digest_pairs' digest from candidate arrays: cand = structured rows (rb, re, qb, qe, score, pos, NM, n_cigar, is_rev,
    csub, seedcov, cigar_off) of consecutive reads, read_off[r] .. read_off[r + 1] = rows of read r (first row of read 0 is row 0),
    cigar = the pool cigar_off points into.  Returns uint64[len(read_off) - 1]."""
    with np.errstate(over="ignore"):
        u = lambda a: np.asarray(a).astype(np.int64).astype(np.uint64)      # noqa: E731 -- sign-extended like the C casts
        m = np.zeros(len(cand), dtype=np.uint64)
        for f, k in zip(("rb", "re", "qb", "qe", "score", "pos", "NM", "n_cigar", "is_rev", "csub", "seedcov"), _DG):
            m += u(cand[f]) * k
        nc = np.asarray(cand["n_cigar"]).astype(np.int64)
        tot = int(nc.sum())
        if tot:
            row = np.repeat(np.arange(len(cand)), nc)
            start = np.cumsum(nc) - nc
            j = np.arange(tot) - np.repeat(start, nc)
            ops = cigar[np.asarray(cand["cigar_off"]).astype(np.int64)[row] + j].astype(np.uint64)
            term = ops * (np.uint64(0xA0761D6478BD642F) + j.astype(np.uint64) * np.uint64(0xE7037ED1A0B428DB))
            np.add.at(m, row, term)
        m ^= m >> np.uint64(29); m *= np.uint64(0x8EBC6AF09C88C6E3); m ^= m >> np.uint64(32)
        read_off = np.asarray(read_off).astype(np.int64)
        n_per = np.diff(read_off)
        r_in = np.arange(len(cand)) - np.repeat(read_off[:-1], n_per)
        m *= (2 * r_in + 1).astype(np.uint64)
        out = np.zeros(len(n_per), dtype=np.uint64)
        np.add.at(out, np.repeat(np.arange(len(n_per)), n_per), m)
        out += n_per.astype(np.uint64) * np.uint64(0x9FB21C651E98DF25)
    return out


# ---- bucket reader oracle (oracle/ingest.c) and the reference's own util.c (libref_util.so, built outside the repository) ----
ORC_MAX_READ_LEN = 255


class FastqRec(C.Structure):
    _fields_ = [("bc", C.c_uint64), ("rlen", C.c_ushort), ("id", C.c_char * 150), ("read", C.c_char * (ORC_MAX_READ_LEN + 2)),
                ("qual", C.c_char * (ORC_MAX_READ_LEN + 2))]


def read_special_fastq(path: str, bc_len: int = 16, is_haplotag: bool = False):
    """[(bc, id, read1, qual1, read2, qual2)] in the reference's order, and the barcode groups [(start, n)]."""
    L = lib()
    L.orc_read_special_fastq.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.POINTER(FastqRec)), C.POINTER(C.POINTER(FastqRec)),
                                         C.POINTER(C.c_size_t)]
    L.orc_next_group.restype = C.c_size_t
    L.orc_next_group.argtypes = [C.POINTER(FastqRec), C.c_size_t]
    r1, r2, n = C.POINTER(FastqRec)(), C.POINTER(FastqRec)(), C.c_size_t()
    assert L.orc_read_special_fastq(path.encode(), bc_len, int(is_haplotag), C.byref(r1), C.byref(r2), C.byref(n)) == 0
    recs = []
    for i in range(n.value):
        a, b = r1[i], r2[i]
        assert a.bc == b.bc and a.id == b.id and a.rlen == len(a.read) and b.rlen == len(b.read)
        recs.append((a.bc, a.id, a.read, a.qual, b.read, b.qual))
    groups, at = [], 0
    while True:
        k = L.orc_next_group(r1, at)
        assert k == L.orc_next_group(r2, at)
        if k == 0:
            break
        groups.append((at, k))
        at += k
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(r1); libc.free(r2)
    return recs, groups


def oracle_encode_bc(bc: bytes, is_haplotag=False) -> int:
    L = lib()
    L.orc_encode_bc.restype = C.c_uint64
    L.orc_encode_bc.argtypes = [C.c_char_p, C.c_int, C.c_int]
    return L.orc_encode_bc(bc, len(bc), int(is_haplotag))


def oracle_decode_bc(v: int, bc_len: int, is_haplotag=False) -> bytes:
    L = lib()
    L.orc_decode_bc.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_char_p]
    buf = C.create_string_buffer(64)
    L.orc_decode_bc(v, bc_len, int(is_haplotag), buf)
    return buf.raw[:12 if is_haplotag else bc_len]


def oracle_copy_until_space(line: bytes, n_fields: int):
    L = lib()
    L.orc_copy_until_space.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
    return _fields_with(L.orc_copy_until_space, line, n_fields)


def _fields_with(fn, line: bytes, n_fields: int):
    buf = C.create_string_buffer(line + b"\0" * 8)      # room for the steps past the terminator
    p = C.cast(buf, C.c_char_p)
    out = []
    for _ in range(n_fields):
        dest = C.create_string_buffer(len(line) + 2)
        fn(dest, C.byref(p))
        out.append(dest.value)
    return out


# Where `make -C oracle ref` puts what it builds from the reference's own sources: OUTSIDE the repository, so that nothing built
# from /root/reference can travel to a GPU box with the working tree (oracle/Makefile, REFOUT).
REF_OUT = os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref")
REF_UTIL = os.path.join(REF_OUT, "libref_util.so")


class RefUtil:
    """The reference's own src/util.c, compiled as it lies (oracle/Makefile `ref`); None-like if it was not built."""
    def __init__(self):
        self.L = C.CDLL(REF_UTIL)
        self.L.encode_bc.restype = C.c_uint64
        self.L.encode_bc.argtypes = [C.c_char_p, C.c_int]
        self.L.decode_bc.argtypes = [C.c_uint64, C.c_char_p, C.c_int]
        self.L.copy_until_space.argtypes = [C.c_char_p, C.POINTER(C.c_char_p)]
        self.L.ref_set_bc_len.argtypes = [C.c_int]

    def encode_bc(self, bc: bytes, is_haplotag=False) -> int:
        self.L.ref_set_bc_len(len(bc))
        return self.L.encode_bc(bc, int(is_haplotag))

    def decode_bc(self, v: int, bc_len: int, is_haplotag=False) -> bytes:
        self.L.ref_set_bc_len(bc_len)
        buf = C.create_string_buffer(64)
        self.L.decode_bc(v, buf, int(is_haplotag))
        return buf.raw[:12 if is_haplotag else bc_len]

    def copy_until_space(self, line: bytes, n_fields: int):
        return _fields_with(self.L.copy_until_space, line, n_fields)


class CRec(C.Structure):      # orc_crec_t (oracle/oracle.h)
    _fields_ = [("bc", C.c_uint64), ("chrom", C.c_uint32), ("pos", C.c_uint32), ("ident", C.c_char * 256), ("score", C.c_double),
                ("mate", C.c_uint32), ("rev", C.c_uint32), ("orig", C.c_uint32), ("hash", C.c_uint32), ("mate_hash", C.c_uint32),
                ("hashed", C.c_uint8), ("mate_hashed", C.c_uint8), ("active", C.c_uint8), ("duplicate", C.c_uint8), ("visited", C.c_uint8),
                ("pad_", C.c_uint8 * 3), ("gamma", C.c_double), ("cloud_id", C.c_int32), ("cloud_bad", C.c_int32), ("alt", C.c_int32),
                ("clip_edit_dist", C.c_int32), ("sel_mate", C.c_void_p)]


def clouds_group(records, n_pairs, cloud_id, dist_thresh=50000, many_clouds=False):
    """oracle/clouds.c on one barcode group.  records: [(bc, chrom, pos, ident, score, mate, rev)] in append_alignments' order.
    Returns (print order [(rec, mate | -1)], per-record results [(gamma, cloud_id, cloud_bad, duplicate, alt)], next cloud id)."""
    L = lib()
    L.orc_clouds_group.restype = C.c_size_t
    L.orc_clouds_group.argtypes = [C.POINTER(CRec), C.c_size_t, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    n = len(records)
    arr = (CRec * max(n, 1))()
    for i, rec_in in enumerate(records):
        bc, chrom, pos, ident, score, mate, rev = rec_in[:7]
        r = arr[i]
        r.bc, r.chrom, r.pos, r.ident, r.score, r.mate, r.rev, r.orig, r.active, r.alt = bc, chrom, pos, ident, score, mate, rev, i, 1, -1
        r.clip_edit_dist = rec_in[7] if len(rec_in) > 7 else 0      # (read by -d only: clouds_density)
    cid = C.c_int(cloud_id)
    order = (C.c_int * (2 * max(n, 1)))()
    k = L.orc_clouds_group(arr, n, n_pairs, dist_thresh, int(many_clouds), C.byref(cid), order)
    out = [(order[2 * i], order[2 * i + 1]) for i in range(k)]
    res = [(arr[i].gamma, arr[i].cloud_id, arr[i].cloud_bad, arr[i].duplicate, arr[i].alt) for i in range(n)]
    return out, res, cid.value


def clouds_density(on: bool, probs=None, seed=None):
    """-d of the oracle's cloud stage (oracle/clouds.c, mark_optimal_alignments_in_cloud): on / off, the platform's density model
    (default: src/techs.c's 0.6, 0.05, 0.2, 0.01), and srand(seed) when a seed is given (the optimiser draws from libc's rand())."""
    L = lib()
    L.orc_clouds_set_density.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double)]
    if probs is None:
        L.orc_clouds_set_density(int(on), 0, None)
    else:
        L.orc_clouds_set_density(int(on), len(probs), (C.c_double * len(probs))(*probs))
    if seed is not None:
        L.orc_clouds_reseed.argtypes = [C.c_uint]
        L.orc_clouds_reseed(seed)


def sam_header(contigs, rg_line, version: bytes, argv) -> bytes:
    """oracle/sam.c's write_sam_header (reference src/align.c:193-212): contigs = [(name, length)]."""
    L = lib()
    L.orc_sam_header.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_char_p, C.c_int,
                                 C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    names = (C.c_char_p * max(1, len(contigs)))(*[n for n, _ in contigs])
    lens = (C.c_int32 * max(1, len(contigs)))(*[l for _, l in contigs])
    av = (C.c_char_p * len(argv))(*argv)
    text, size = C.c_void_p(), C.c_size_t()
    assert L.orc_sam_header(names, lens, len(contigs), rg_line, version, len(argv), av, C.byref(text), C.byref(size)) == 0
    out = C.string_at(text, size.value)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(text)
    return out


class _ChainV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


def n_chains(idx: Index, opt, read_ascii: bytes) -> int:
    """Chains mem_chain builds for a read, before the filter (orc_chain): how chain-rich a test read is."""
    L = lib()
    L.orc_chain.restype = _ChainV
    L.orc_chain.argtypes = [C.POINTER(Opt), C.c_void_p, C.c_int, C.c_char_p]
    nt4 = bytes({65: 0, 67: 1, 71: 2, 84: 3}.get(c, 4) for c in read_ascii)
    v = L.orc_chain(C.byref(opt), idx.h, len(nt4), nt4)
    return int(v.n)      # (the few KB of seeds are left to the process: a test helper)
