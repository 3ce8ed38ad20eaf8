"""ctypes mirror of include/ema_bwaabi.h (the libbwa link surface, libema_bwaabi.so) for the tests, and the reference's
bridge replayed on top of it: bwa_mem_mate_sw + bwa_smith_waterman exactly as reference src/bwabridge.c:204-311 calls the
nine symbols."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class MemOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("a", "b", "o_del", "e_del", "o_ins", "e_ins", "pen_unpaired", "pen_clip5", "pen_clip3", "w", "zdrop")] + \
               [("max_mem_intv", C.c_uint64)] + \
               [(n, C.c_int) for n in ("T", "flag", "min_seed_len", "min_chain_weight", "max_chain_extend")] + [("split_factor", C.c_float)] + \
               [(n, C.c_int) for n in ("split_width", "max_occ", "max_chain_gap", "n_threads", "chunk_size")] + \
               [(n, C.c_float) for n in ("mask_level", "drop_ratio", "XA_drop_ratio", "mask_level_redun", "mapQ_coef_len")] + \
               [(n, C.c_int) for n in ("mapQ_coef_fac", "max_ins", "max_matesw", "max_XA_hits", "max_XA_hits_alt")] + [("mat", C.c_int8 * 25)]


class AlnReg(C.Structure):
    _fields_ = [("rb", C.c_int64), ("re", C.c_int64)] + \
               [(n, C.c_int) for n in ("qb", "qe", "rid", "score", "truesc", "sub", "alt_sc", "csub", "sub_n", "w", "seedcov", "secondary",
                                       "secondary_all", "seedlen0")] + \
               [("n_comp", C.c_int, 30), ("is_alt", C.c_int, 2), ("frac_rep", C.c_float), ("hash", C.c_uint64)]


class AlnRegV(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.POINTER(AlnReg))]


class PeStat(C.Structure):
    _fields_ = [("low", C.c_int), ("high", C.c_int), ("failed", C.c_int), ("avg", C.c_double), ("std", C.c_double)]


class Aln(C.Structure):      # with the bit-field read as one word, as the reference's cast does (src/bwabridge.c:159-168)
    _fields_ = [("pos", C.c_int64), ("rid", C.c_int), ("flag", C.c_int), ("flag2", C.c_uint32), ("n_cigar", C.c_int),
                ("cigar", C.POINTER(C.c_uint32)), ("XA", C.c_char_p), ("score", C.c_int), ("sub", C.c_int), ("alt_sc", C.c_int)]


class Ann(C.Structure):
    _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("n_ambs", C.c_int32), ("gi", C.c_uint32), ("is_alt", C.c_int32),
                ("name", C.c_char_p), ("anno", C.c_char_p)]


class BntSeq(C.Structure):
    _fields_ = [("l_pac", C.c_int64), ("n_seqs", C.c_int32), ("seed", C.c_uint32), ("anns", C.POINTER(Ann)), ("n_holes", C.c_int32),
                ("ambs", C.c_void_p), ("fp_pac", C.c_void_p)]


class BwaIdx(C.Structure):
    _fields_ = [("bwt", C.c_void_p), ("bns", C.POINTER(BntSeq)), ("pac", C.POINTER(C.c_uint8)), ("is_shm", C.c_int), ("l_mem", C.c_int64),
                ("mem", C.c_void_p)]


SIZES = {0: C.sizeof(MemOpt), 1: C.sizeof(AlnReg), 2: C.sizeof(Aln), 3: C.sizeof(PeStat), 4: C.sizeof(Ann), 5: C.sizeof(BntSeq), 6: C.sizeof(BwaIdx)}
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "ema_amd", "libema_bwaabi.so"))
        L.bwa_idx_load.restype = C.POINTER(BwaIdx)
        L.bwa_idx_load.argtypes = [C.c_char_p, C.c_int]
        L.bwa_idx_destroy.argtypes = [C.POINTER(BwaIdx)]
        L.mem_opt_init.restype = C.POINTER(MemOpt)
        L.mem_align1_core.restype = AlnRegV
        L.mem_align1_core.argtypes = [C.POINTER(MemOpt), C.c_void_p, C.POINTER(BntSeq), C.POINTER(C.c_uint8), C.c_int, C.c_char_p, C.c_void_p]
        L.mem_matesw.restype = C.c_int
        L.mem_matesw.argtypes = [C.POINTER(MemOpt), C.POINTER(BntSeq), C.POINTER(C.c_uint8), C.POINTER(PeStat), C.POINTER(AlnReg), C.c_int,
                                 C.c_char_p, C.POINTER(AlnRegV)]
        L.mem_reg2aln.restype = Aln
        L.mem_reg2aln.argtypes = [C.POINTER(MemOpt), C.POINTER(BntSeq), C.POINTER(C.c_uint8), C.c_int, C.c_char_p, C.POINTER(AlnReg)]
        L.bns_fetch_seq.restype = C.POINTER(C.c_uint8)
        L.bns_fetch_seq.argtypes = [C.POINTER(BntSeq), C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int)]
        L.ema_bwaabi_sizeof.restype = C.c_size_t
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def nt4(read: bytes) -> bytes:
    t = (C.c_ubyte * 256).in_dll(lib(), "nst_nt4_table")
    return bytes(t[c] for c in read)      # seq_convert, reference src/bwabridge.c:151-157


def bridge_pair(idx, opt, read1: bytes, read2: bytes, score_delta=25):
    """bwa_mem_mate_sw (reference src/bwabridge.c:204-299) then bwa_smith_waterman for every hit (:301-311, called at
    src/align.c:1013,1038), through the nine symbols only.  Returns per mate a list of dicts like the oracle's align_pair."""
    L = lib()
    pes = (PeStat * 4)()
    for i in range(4):
        pes[i].failed = 0 if i == 1 else 1
        pes[i].low, pes[i].high, pes[i].avg, pes[i].std = -35, 500, 200.0, 100.0
    ix = idx.contents
    s1, s2 = C.create_string_buffer(nt4(read1), len(read1)), C.create_string_buffer(nt4(read2), len(read2))
    r1 = L.mem_align1_core(opt, ix.bwt, ix.bns, ix.pac, len(read1), s1, None)
    r2 = L.mem_align1_core(opt, ix.bwt, ix.bns, ix.pac, len(read2), s2, None)
    best1 = max([0] + [r1.a[i].score for i in range(r1.n)])
    best2 = max([0] + [r2.a[i].score for i in range(r2.n)])
    num = 0
    i = 0
    while i < r2.n and num < 50:
        if r2.a[i].score >= best2 - score_delta:
            num += 1
            L.mem_matesw(opt, ix.bns, ix.pac, pes, C.byref(r2.a[i]), len(read1), s1, C.byref(r1))
        i += 1
    num = 0
    i = 0
    while i < r1.n and num < 50:
        if r1.a[i].score >= best1 - score_delta:      # best1: the pre-rescue best, as in the reference
            num += 1
            anchor = AlnReg.from_buffer_copy(r1.a[i])
            L.mem_matesw(opt, ix.bns, ix.pac, pes, C.byref(anchor), len(read2), s2, C.byref(r2))
        i += 1
    out = []
    for rv, s, n in ((r1, s1, len(read1)), (r2, s2, len(read2))):
        hits = []
        for k in range(rv.n):
            g = rv.a[k]
            a = L.mem_reg2aln(opt, ix.bns, ix.pac, n, s, C.byref(g))
            d = dict(rb=g.rb, re=g.re, qb=g.qb, qe=g.qe, rid=g.rid, score=g.score, truesc=g.truesc, sub=g.sub, csub=g.csub, w=g.w,
                     seedcov=g.seedcov, secondary=g.secondary, seedlen0=g.seedlen0, n_comp=g.n_comp, is_alt=g.is_alt, frac_rep=float(g.frac_rep),
                     pos=a.pos, is_rev=a.flag2 & 1, NM=a.flag2 >> 10, mapq=(a.flag2 & 0x3fc) >> 2, cigar=[a.cigar[j] for j in range(a.n_cigar)])
            _libc.free(a.cigar)
            hits.append(d)
        out.append(hits)
    _libc.free(r1.a); _libc.free(r2.a)
    return out
