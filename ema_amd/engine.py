"""ctypes binding of the engine's C ABI (include/ema_engine.h) for the tests and bench.py.

The product boundary is the C ABI itself; this module only marshals numpy arrays across it.
It never computes alignments on the CPU: if `libema_engine.so` or a GPU is missing, it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# EMA_ENGINE_LIB: another in-tree build of the same sources (tests: libema_engine_ss16.so, small rank superblocks)
LIB_PATH = os.path.join(_HERE, os.environ.get("EMA_ENGINE_LIB", "libema_engine.so"))

SYMBOLS = [
    "ema_engine_opts_default", "ema_engine_open", "ema_engine_close", "ema_engine_strerror", "ema_engine_n_contigs",
    "ema_engine_contig_name", "ema_engine_contig_len", "ema_engine_contig_offset", "ema_engine_contig_is_alt", "ema_engine_l_pac",
    "ema_engine_align_pairs", "ema_batch_free", "ema_engine_batch_capacity", "ema_engine_stage", "ema_engine_run",
    "ema_engine_sync", "ema_engine_fetch", "ema_engine_debug_seeds", "ema_engine_last_timing", "ema_engine_debug_dp", "ema_engine_debug_regions", "ema_engine_debug_dedup", "ema_engine_debug_contigs", "ema_engine_n_streams", "ema_engine_full_tier_capacity", "ema_engine_run_serial", "ema_batch_append_alignments", "ema_aln_free", "ema_engine_open_shared", "ema_engine_index_info", "ema_engine_stage_slot", "ema_engine_run_slot", "ema_engine_peer",
    "ema_engine_seed_launches", "ema_engine_get_opts", "ema_engine_debug_sa",
]


class Opts(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_clip5", "pen_clip3", "w",
                                       "zdrop", "min_seed_len", "split_width", "max_mem_intv", "max_occ",
                                       "max_chain_gap", "min_chain_weight", "max_chain_extend")] + \
               [(n, C.c_float) for n in ("split_factor", "mask_level", "drop_ratio", "mask_level_redun")] + \
               [(n, C.c_int) for n in ("score_delta", "max_rescue", "pes_low", "pes_high", "batch_pairs", "n_streams", "full_tier_pairs", "lean_intervals", "lean_regions", "lean_cigar_ops", "mapq_coef_len", "mapq_coef_fac", "lean_seed_extends")]


class Cand(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64)] + \
               [(n, C.c_int32) for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
                                         "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] + \
               [("frac_rep", C.c_float), ("pos", C.c_int64)] + \
               [(n, C.c_int32) for n in ("is_rev", "NM", "n_cigar")] + [("cigar_off", C.c_uint32)] + \
               [(n, C.c_int32) for n in ("aln_score", "aln_sub")]


CAND_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8")] +
                      [(n, "<i4") for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w",
                                            "seedcov", "secondary", "secondary_all", "seedlen0", "n_comp", "is_alt")] +
                      [("frac_rep", "<f4"), ("pos", "<i8")] +
                      [(n, "<i4") for n in ("is_rev", "NM", "n_cigar")] + [("cigar_off", "<u4")] +
                      [(n, "<i4") for n in ("aln_score", "aln_sub")], align=True)


REG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8")] +
                     [(n, "<i4") for n in ("qb", "qe", "rid", "score", "truesc", "sub", "csub", "w", "seedcov", "secondary",
                                           "seedlen0", "n_comp", "is_alt")] + [("frac_rep", "<f4")], align=True)


class BatchOut(C.Structure):
    _fields_ = [("n_pairs", C.c_size_t), ("cand_off", C.POINTER(C.c_uint64)), ("cand", C.POINTER(Cand)),
                ("cigar", C.POINTER(C.c_uint32)), ("n_cigar", C.c_size_t), ("n_redone", C.c_size_t), ("status", C.POINTER(C.c_int32)),
                ("redone", C.POINTER(C.c_uint32)), ("view_of", C.c_void_p)]


class AlnRec(C.Structure):
    _fields_ = [("pair", C.c_uint32), ("mate", C.c_uint8), ("unique", C.c_uint8), ("pad_", C.c_uint8 * 2), ("cand", C.c_uint64),
                ("clip", C.c_int32), ("clip_edit_dist", C.c_int32), ("mapq", C.c_int32), ("score_mapq", C.c_int32),
                ("score", C.c_double)]


ALN_REC_DTYPE = np.dtype([("pair", "<u4"), ("mate", "u1"), ("unique", "u1"), ("pad_", "u1", (2,)), ("cand", "<u8"), ("clip", "<i4"),
                          ("clip_edit_dist", "<i4"), ("mapq", "<i4"), ("score_mapq", "<i4"), ("score", "<f8")], align=True)


class AlnOut(C.Structure):
    _fields_ = [("n_pairs", C.c_size_t), ("n", C.c_size_t), ("pair_off", C.POINTER(C.c_uint64)), ("rec", C.POINTER(AlnRec))]


class Timing(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("seed_ms", "chain_ms", "extend_ms", "rescue_ms", "final_ms", "total_ms", "full_tier_ms")] + [("full_ms", C.c_float * 4)]


_lib = None
# one hardware queue per batch slice + the full-capacity tier (read by the ROCm runtime when it initialises; see engine.hip)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def load_library():
    """Loads the in-tree HIP library.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make` or __graft_entry__.build(); "
                               "the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        L.ema_engine_opts_default.argtypes = [C.POINTER(Opts)]
        L.ema_engine_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(Opts), C.POINTER(C.c_void_p)]
        L.ema_engine_close.argtypes = [C.c_void_p]
        L.ema_engine_strerror.restype = C.c_char_p
        L.ema_engine_strerror.argtypes = [C.c_void_p]
        L.ema_engine_n_contigs.argtypes = [C.c_void_p]
        L.ema_engine_contig_name.restype = C.c_char_p
        L.ema_engine_contig_name.argtypes = [C.c_void_p, C.c_int]
        L.ema_engine_contig_len.restype = C.c_int64
        L.ema_engine_contig_len.argtypes = [C.c_void_p, C.c_int]
        L.ema_engine_contig_offset.restype = C.c_int64
        L.ema_engine_contig_offset.argtypes = [C.c_void_p, C.c_int]
        L.ema_engine_l_pac.restype = C.c_int64
        L.ema_engine_l_pac.argtypes = [C.c_void_p]
        L.ema_engine_batch_capacity.restype = C.c_size_t
        L.ema_engine_batch_capacity.argtypes = [C.c_void_p]
        L.ema_engine_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.ema_engine_run.argtypes = [C.c_void_p]
        L.ema_engine_run_serial.argtypes = [C.c_void_p]
        L.ema_engine_sync.argtypes = [C.c_void_p]
        L.ema_engine_fetch.argtypes = [C.c_void_p, C.POINTER(C.POINTER(BatchOut))]
        L.ema_engine_align_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.POINTER(C.POINTER(BatchOut))]
        L.ema_batch_free.argtypes = [C.POINTER(BatchOut)]
        L.ema_engine_debug_seeds.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint64)),
                                             C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]
        L.ema_engine_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        L.ema_engine_n_streams.argtypes = [C.c_void_p]
        L.ema_engine_full_tier_capacity.argtypes = [C.c_void_p]
        L.ema_engine_full_tier_capacity.restype = C.c_size_t
        L.ema_engine_debug_regions.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.POINTER(C.c_int32)),
                                               C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_int32)]
        L.ema_engine_debug_dedup.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ema_engine_debug_contigs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ema_engine_debug_dp.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.ema_engine_set_tuning.argtypes = [C.c_char_p]
        _lib = L
        tuning_from_env(LEGACY_KNOB_VARS)      # (the A/B scripts under tools/ still say EMA_SEED_TAIL=0 python3 ...: translated here, once)
    return _lib


# The knobs that were environment variables of the LIBRARY up to round 4; the library itself now reads only EMA_TUNING / set_tuning.
LEGACY_KNOB_VARS = ("EMA_KMER_K", "EMA_SEED_TAIL", "EMA_SEED_WTEST", "EMA_SEED_ANCHOR", "EMA_SEED_ONEPASS", "EMA_SEED_ROUNDS", "EMA_SEED_PARK",
                    "EMA_SEED_BLOCKS_PER_CU", "EMA_SEED_ORDER", "EMA_SEED_LONG_WAVE", "EMA_FULL_SEED_LANE", "EMA_FULL_OWN_STREAM", "EMA_GRID",
                    "EMA_DEVICE_MERGE", "EMA_LANE_ALIGN", "EMA_HEAVY_CHAINS", "EMA_HEAVY_ATTEMPTS", "EMA_HEAVY_REGIONS", "EMA_SMALL_ONE_SLICE",
                    "EMA_LEAN_INTERVALS", "EMA_LEAN_REGIONS", "EMA_PHASE_PROFILE", "EMA_WATCHDOG_S", "EMA_WATCHDOG_NOMARK", "EMA_DP_TIMING",
                    "EMA_ALIGN_PIPELINE", "EMA_VERBOSE")


_tuning = {}


def set_tuning(**knobs):
    """Development knobs of the library (include/ema_engine.h, ema_engine_set_tuning): `set_tuning(seed_tail=0, kmer_k=12)`;
    a value of None removes a knob; no arguments at all clears them (back to the EMA_TUNING environment variable, if any).
    Read when an engine is opened.  This is what the tests and tools use instead of one environment variable per knob."""
    L = _lib if _lib is not None else load_library()
    if not knobs:
        _tuning.clear()
    for k, v in knobs.items():
        if v is None:
            _tuning.pop(k, None)
        else:
            _tuning[k] = str(v).replace(",", ":")      # ',' separates knobs in the library's string: a knob's own list ("grid", "seed_order") uses ':'
    L.ema_engine_set_tuning(",".join(f"{k}={v}" for k, v in _tuning.items()).encode() if _tuning else None)


def tuning_from_env(names, environ=None):
    """For command-line tools that take A/B knobs from their environment (tools/*.sh): EMA_<KNOB> -> knob, for the names given."""
    environ = os.environ if environ is None else environ
    got = {n[4:].lower(): environ[n] for n in names if n in environ}
    if got:
        set_tuning(**got)
    return got


def default_opts() -> Opts:
    o = Opts()
    load_library().ema_engine_opts_default(C.byref(o))
    return o


class Batch:
    """Result of one batch, copied out of the engine's buffers."""

    def __init__(self, cand_off, cand, cigar, status, n_redone=0, redone=None):
        self.cand_off, self.cand, self.cigar, self.status = cand_off, cand, cigar, status
        self.n_redone = n_redone      # pairs that went through the full-capacity tier
        self.redone = redone if redone is not None else np.zeros(0, np.uint32)

    def mate(self, pair, m):
        lo, hi = int(self.cand_off[2 * pair + m]), int(self.cand_off[2 * pair + m + 1])
        return self.cand[lo:hi]

    def cigar_of(self, c):
        return self.cigar[int(c["cigar_off"]):int(c["cigar_off"]) + int(c["n_cigar"])]


def gather_pairs(ob, pair_ids):
    """Candidates of the given pairs out of an ema_batch_out (`ob`: the ctypes struct a stream sink receives, or a Batch): copies
    of the candidate rows of reads 2p, 2p + 1 for every p in pair_ids, in that order, with cigar_off rebased into the returned
    pool.  Returns (cand rows, CIGAR pool, read_off[2 * len(pair_ids) + 1]).  Vectorised: cheap enough for a sink callback."""
    ids = np.asarray(pair_ids, dtype=np.int64)
    if isinstance(ob, Batch):
        cand_off, cand_all, cig_all = ob.cand_off, ob.cand, ob.cigar
    else:
        n = int(ob.n_pairs)
        cand_off = np.ctypeslib.as_array(ob.cand_off, shape=(2 * n + 1,))
        n_cand = int(cand_off[-1])
        cand_all = np.frombuffer((C.c_char * (n_cand * C.sizeof(Cand))).from_address(C.addressof(ob.cand.contents)), dtype=CAND_DTYPE) \
            if n_cand else np.zeros(0, dtype=CAND_DTYPE)
        cig_all = np.ctypeslib.as_array(ob.cigar, shape=(max(int(ob.n_cigar), 1),))
    reads = np.stack([2 * ids, 2 * ids + 1], axis=1).reshape(-1)
    lo = cand_off[reads].astype(np.int64)
    cnt = cand_off[reads + 1].astype(np.int64) - lo
    read_off = np.zeros(len(reads) + 1, dtype=np.int64)
    read_off[1:] = np.cumsum(cnt)
    tot = int(read_off[-1])
    rows = np.repeat(lo - read_off[:-1], cnt) + np.arange(tot)
    cand = cand_all[rows].copy() if tot else np.zeros(0, dtype=CAND_DTYPE)
    nc = cand["n_cigar"].astype(np.int64)
    c_off = np.zeros(tot + 1, dtype=np.int64)
    c_off[1:] = np.cumsum(nc)
    n_ops = int(c_off[-1])
    src = np.repeat(cand["cigar_off"].astype(np.int64) - c_off[:-1], nc) + np.arange(n_ops)
    pool = cig_all[src].copy() if n_ops else np.zeros(0, dtype=np.uint32)
    cand["cigar_off"] = c_off[:-1].astype(np.uint32)
    return cand, pool, read_off


def append_alignments(batch: "Batch", off: np.ndarray, opts: "Opts | None" = None, error_rate: float = 0.001):
    """The reference's append_alignments() on a batch (reference src/align.c:986-1061; host arithmetic, no GPU): returns
    (records as a structured array in the reference's order, pair_off[n_pairs + 1])."""
    L = load_library()
    L.ema_batch_append_alignments.argtypes = [C.POINTER(BatchOut), C.c_void_p, C.POINTER(Opts), C.c_double, C.POINTER(C.POINTER(AlnOut))]
    L.ema_aln_free.argtypes = [C.POINTER(AlnOut)]
    o = opts if opts is not None else default_opts()
    off = np.ascontiguousarray(off, dtype=np.uint32)
    cand_off = np.ascontiguousarray(batch.cand_off, dtype=np.uint64)
    cand = np.ascontiguousarray(batch.cand)
    cigar = np.ascontiguousarray(batch.cigar, dtype=np.uint32)
    status = np.ascontiguousarray(batch.status, dtype=np.int32)
    b = BatchOut((len(cand_off) - 1) // 2, cand_off.ctypes.data_as(C.POINTER(C.c_uint64)), cand.ctypes.data_as(C.POINTER(Cand)),
                 cigar.ctypes.data_as(C.POINTER(C.c_uint32)), len(cigar), batch.n_redone, status.ctypes.data_as(C.POINTER(C.c_int32)), None)
    p = C.POINTER(AlnOut)()
    rc = L.ema_batch_append_alignments(C.byref(b), off.ctypes.data, C.byref(o), error_rate, C.byref(p))
    if rc != 0:
        raise RuntimeError(f"ema_batch_append_alignments failed ({rc})")
    try:
        a = p.contents
        rec = np.frombuffer(C.string_at(a.rec, a.n * C.sizeof(AlnRec)), dtype=ALN_REC_DTYPE).copy() if a.n else np.zeros(0, ALN_REC_DTYPE)
        pair_off = np.ctypeslib.as_array(a.pair_off, shape=(a.n_pairs + 1,)).copy()
    finally:
        L.ema_aln_free(p)
    return rec, pair_off


class Engine:
    """Mirror of the reference's bridge for a batch of pairs (reference include/bwabridge.h:92-106):
    `Engine(prefix)` ~ load_reference(); `align_pairs()` ~ bwa_mem_mate_sw + bwa_smith_waterman for every
    candidate of every pair."""

    def __init__(self, index_prefix: str | None, device: int = 0, opts: Opts | None = None, share: "Engine | None" = None):
        """share: another Engine on the same GPU whose index this one uses (own buffers and streams; see ema_engine_open_shared)."""
        self._L = load_library()
        self._h = C.c_void_p()
        if share is not None:
            self._L.ema_engine_open_shared.argtypes = [C.c_void_p, C.POINTER(Opts), C.POINTER(C.c_void_p)]
            rc = self._L.ema_engine_open_shared(share._h, C.byref(opts) if opts is not None else None, C.byref(self._h))
            if rc != 0:
                raise RuntimeError(f"ema_engine_open_shared failed ({rc}): {self._L.ema_engine_strerror(self._h).decode()}")
            self._n_reads_staged = 0
            return
        rc = self._L.ema_engine_open(index_prefix.encode(), device, C.byref(opts) if opts is not None else None,
                                     C.byref(self._h))
        if rc != 0:
            msg = self._L.ema_engine_strerror(self._h).decode() if self._h else "allocation failure"
            if self._h:
                self._L.ema_engine_close(self._h)
                self._h = C.c_void_p()
            raise RuntimeError(f"ema_engine_open failed ({rc}): {msg}")

    _borrowed = False      # a peer handle belongs to its engine

    def close(self):
        if self._h and not self._borrowed:
            self._L.ema_engine_close(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._L.ema_engine_strerror(self._h).decode()}")

    @property
    def n_streams(self):
        return int(self._L.ema_engine_n_streams(self._h))

    @property
    def full_tier_capacity(self):
        return int(self._L.ema_engine_full_tier_capacity(self._h))

    @property
    def capacity(self):
        return int(self._L.ema_engine_batch_capacity(self._h))

    def contigs(self):
        n = self._L.ema_engine_n_contigs(self._h)
        return [(self._L.ema_engine_contig_name(self._h, i).decode(), int(self._L.ema_engine_contig_len(self._h, i)),
                 int(self._L.ema_engine_contig_offset(self._h, i))) for i in range(n)]

    def stage(self, bases: np.ndarray, off: np.ndarray):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        self._check(self._L.ema_engine_stage(self._h, bases.ctypes.data, off.ctypes.data, (len(off) - 1) // 2), "stage")
        self._n_reads_staged = len(off) - 1

    def stage_slot(self, slot: int, bases: np.ndarray, off: np.ndarray):
        """ema_engine_stage_slot: a batch into input slot `slot` (several batches resident at once)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        self._L.ema_engine_stage_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        self._check(self._L.ema_engine_stage_slot(self._h, slot, bases.ctypes.data, off.ctypes.data, (len(off) - 1) // 2), "stage_slot")
        self._n_reads_staged = len(off) - 1

    def run_slot(self, slot: int):
        self._L.ema_engine_run_slot.argtypes = [C.c_void_p, C.c_int]
        self._check(self._L.ema_engine_run_slot(self._h, slot), "run_slot")

    def peer(self) -> "Engine | None":
        """The engine's own second set of batch buffers (ema_engine_peer), owned by this engine; None if it cannot be created."""
        self._L.ema_engine_peer.restype = C.c_void_p
        self._L.ema_engine_peer.argtypes = [C.c_void_p]
        h = self._L.ema_engine_peer(self._h)
        if not h:
            return None
        p = Engine.__new__(Engine)
        p._L, p._h, p._borrowed = self._L, C.c_void_p(h), True
        return p

    def seed_launches_per_series(self) -> int:
        self._L.ema_engine_seed_launches.argtypes = [C.c_void_p]
        return int(self._L.ema_engine_seed_launches(self._h))

    def run(self, serial: bool = False):
        """Queue one pass over the staged batch (asynchronous).  serial=True: slices one after another, for isolated kernel times."""
        self._check((self._L.ema_engine_run_serial if serial else self._L.ema_engine_run)(self._h), "run")

    def sync(self):
        self._check(self._L.ema_engine_sync(self._h), "sync")

    def timing(self):
        t = Timing()
        self._check(self._L.ema_engine_last_timing(self._h, C.byref(t)), "timing")
        d = {n: getattr(t, n) for n, _ in Timing._fields_ if n != "full_ms"}
        d["full_ms"] = [float(x) for x in t.full_ms]
        return d

    def fetch(self, allow_limit: bool = False) -> Batch:
        p = C.POINTER(BatchOut)()
        rc = self._L.ema_engine_fetch(self._h, C.byref(p))
        if rc != 0 and not (rc == -4 and allow_limit and p):      # EMA_ELIMIT still returns the batch with its status flags
            if p:
                self._L.ema_batch_free(p)
            self._check(rc, "fetch")
        return self._take(p)

    def _take(self, p) -> Batch:
        """Copies an ema_batch_out into numpy arrays and frees it."""
        try:
            o = p.contents
            n = o.n_pairs
            cand_off = np.ctypeslib.as_array(o.cand_off, shape=(2 * n + 1,)).copy()
            n_cand = int(cand_off[-1])
            cand = np.frombuffer(C.string_at(o.cand, n_cand * C.sizeof(Cand)), dtype=CAND_DTYPE).copy() \
                if n_cand else np.zeros(0, dtype=CAND_DTYPE)
            cigar = np.ctypeslib.as_array(o.cigar, shape=(max(o.n_cigar, 1),)).copy()[:o.n_cigar]
            status = np.ctypeslib.as_array(o.status, shape=(2 * n,)).copy() if n else np.zeros(0, np.int32)
            n_redone = int(o.n_redone)
            redone = np.ctypeslib.as_array(o.redone, shape=(max(n_redone, 1),)).copy()[:n_redone]
        finally:
            self._L.ema_batch_free(p)
        return Batch(cand_off, cand, cigar, status, n_redone, redone)

    def align_pairs_any(self, bases: np.ndarray, off: np.ndarray) -> Batch:
        """ema_engine_align_pairs itself (one C call; any number of pairs, worked through in capacity-sized pieces)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint32)
        p = C.POINTER(BatchOut)()
        self._L.ema_engine_align_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(BatchOut))]
        rc = self._L.ema_engine_align_pairs(self._h, bases.ctypes.data, off.ctypes.data, (len(off) - 1) // 2, C.byref(p))
        if rc != 0:
            if p:
                self._L.ema_batch_free(p)
            self._check(rc, "align_pairs")
        return self._take(p)

    def align_pairs(self, bases: np.ndarray, off: np.ndarray) -> Batch:
        self.stage(bases, off)
        self.run()
        self.sync()
        return self.fetch()

    def index_info(self):
        """Layout of the index in HBM (ema_engine_index_info)."""
        a = (C.c_int32 * 4)()
        self._L.ema_engine_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._check(self._L.ema_engine_index_info(self._h, a), "index_info")
        return {"n_super": a[0], "super_shift": a[1], "sa_width": a[2], "kmer_k": a[3]}

    def debug_grids(self):
        """Blocks per compute unit of the kernels' launches (ema_engine_debug_grids)."""
        a = (C.c_int32 * 5)()
        self._L.ema_engine_debug_grids.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._check(self._L.ema_engine_debug_grids(self._h, a), "debug_grids")
        return dict(zip(("k1", "k2a", "k2b", "k3", "k4"), (int(x) for x in a)))

    def debug_sa(self, first: int, n: int) -> np.ndarray:
        """Rows [first, first + n) of the suffix array in HBM (ema_engine_debug_sa)."""
        out = np.zeros(n, dtype=np.uint64)
        self._L.ema_engine_debug_sa.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
        self._check(self._L.ema_engine_debug_sa(self._h, first, n, out.ctypes.data), "debug_sa")
        return out

    def debug_seeds(self):
        """Seed intervals of the staged batch: (intv[n_reads, cap, 4] u64 = k, k', size, start<<32|end; n_intv)."""
        pi, pn, cap = C.POINTER(C.c_uint64)(), C.POINTER(C.c_int32)(), C.c_int32()
        self._check(self._L.ema_engine_debug_seeds(self._h, C.byref(pi), C.byref(pn), C.byref(cap)), "debug_seeds")
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        try:
            n_reads = self._n_reads_staged
            intv = np.ctypeslib.as_array(pi, shape=(n_reads, cap.value, 4)).copy()
            n_intv = np.ctypeslib.as_array(pn, shape=(n_reads,)).copy()
        finally:
            libc.free(pi)
            libc.free(pn)
        return intv, n_intv

    _n_reads_staged = 0

    def debug_regions(self):
        """Regions of the staged reads before mate rescue: (regs[n_reads, cap] structured, n_regs, status)."""
        pr, pn, ps, cap, nb = C.c_void_p(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.c_int32(), C.c_int32()
        self._check(self._L.ema_engine_debug_regions(self._h, C.byref(pr), C.byref(pn), C.byref(ps), C.byref(cap),
                                                     C.byref(nb)), "debug_regions")
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        try:
            n = self._n_reads_staged
            assert nb.value == REG_DTYPE.itemsize
            regs = np.frombuffer(C.string_at(pr, n * cap.value * nb.value), dtype=REG_DTYPE).reshape(n, cap.value).copy()
            n_regs = np.ctypeslib.as_array(pn, shape=(n,)).copy()
            status = np.ctypeslib.as_array(ps, shape=(n,)).copy()
        finally:
            libc.free(pr)
            libc.free(pn)
            libc.free(ps)
        return regs, n_regs, status

    def debug_contigs(self, ctg_off, rb, re):
        """bns_intv2rid of [rb, re) and bns_pos2rid of rb on the device, for a contig layout of the caller's (ctg_off: n + 1 offsets,
        the last one l_pac).  Returns (intv2rid, pos2rid)."""
        ctg_off = np.ascontiguousarray(ctg_off, dtype=np.int64)
        rb = np.ascontiguousarray(rb, dtype=np.int64)
        re = np.ascontiguousarray(re, dtype=np.int64)
        out = np.zeros((len(rb), 2), dtype=np.int32)
        self._check(self._L.ema_engine_debug_contigs(self._h, ctg_off.ctypes.data, len(ctg_off) - 1, rb.ctypes.data, re.ctypes.data,
                                                     len(rb), out.ctypes.data), "debug_contigs")
        return out[:, 0].copy(), out[:, 1].copy()

    def debug_dedup(self, regs, n_in):
        """mem_sort_dedup_patch (no patching) on regs[n_tasks, cap] (REG_DTYPE); returns (regs, n_out)."""
        regs = np.ascontiguousarray(regs.copy())
        n_in = np.ascontiguousarray(n_in, dtype=np.int32)
        n_out = np.zeros(len(n_in), dtype=np.int32)
        self._check(self._L.ema_engine_debug_dedup(self._h, regs.ctypes.data, n_in.ctypes.data, n_out.ctypes.data,
                                                   regs.shape[1], len(n_in)), "debug_dedup")
        return regs, n_out

    def debug_dp(self, kind, qbuf, qoff, tbuf, toff, prm, cigar_cap=512):
        """Runs one of the wave DPs (0 extend, 1 global, 2 local pass) on n tasks; returns (out, cigar|None)."""
        n = len(qoff) - 1
        n_out = {0: 6, 1: 2, 2: 5}[kind]
        out = np.zeros((n, n_out), dtype=np.int32)
        cig = np.zeros((n, cigar_cap), dtype=np.uint32) if kind == 1 else None
        prm = np.ascontiguousarray(prm, dtype=np.int32)
        self._check(self._L.ema_engine_debug_dp(self._h, kind, qbuf.ctypes.data, qoff.ctypes.data, tbuf.ctypes.data,
                                                toff.ctypes.data, prm.ctypes.data, n, out.ctypes.data,
                                                cig.ctypes.data if cig is not None else None, cigar_cap), "debug_dp")
        return out, cig

    def stage_pairs(self, pairs):
        self.stage(pairs.bases, pairs.off)
