"""SAM record formatter (include/ema_sam.h, SURVEY 8f rank 1, writer part): the product's batch formatter against the
oracle's stdio restatement of print_sam_record (reference src/samrecord.c:104-284) on the same records: every
combination of mapped / unmapped record and mate, both strands, duplicates, proper and improper pairs (including the
reference's unsigned position difference), gamma values around the MAPQ thresholds, hard and soft clips, XA lists, 10x
and haplotag barcodes, RG ids with trailing text.  CPU only: the formatter is host code."""
import ctypes as C
import random

import pytest

import oracle_lib as O
from ema_amd import sam

CHROMS = [b"chr1", b"chr2", b"chrX", b"chrUn_KI270742v1"]


class Pool:
    """Keeps every buffer the ctypes structs point at alive."""
    def __init__(self):
        self.keep = []

    def cigar(self, ops):
        a = (C.c_uint32 * max(1, len(ops)))(*[n << 4 | t for n, t in ops])
        self.keep.append(a)
        return a


def rand_cigar(rng, read_len):
    ops, left = [], read_len
    if rng.random() < 0.3:
        k = rng.randrange(1, 20); ops.append((k, rng.choice((3, 4)))); left -= k
    tail = None
    if rng.random() < 0.3:
        k = rng.randrange(1, 20); tail = (k, rng.choice((3, 4))); left -= k
    while left > 0:
        k = rng.randrange(1, left + 1)
        ops.append((k, 0)); left -= k
        if left > 0 and rng.random() < 0.5:
            if rng.random() < 0.5:
                ops.append((rng.randrange(1, 6), 2))
            else:
                j = rng.randrange(1, min(5, left) + 1); ops.append((j, 1)); left -= j
    if tail:
        ops.append(tail)
    return ops


def make_record(rng, pool, ident, bc, mate_no, haplotag):
    r = sam.SamRec()
    l1, l2 = rng.randrange(30, 151), rng.randrange(30, 151)
    chrom_id = rng.randrange(len(CHROMS))
    ops = rand_cigar(rng, l1) if rng.random() < 0.95 else []
    r.ident, r.chrom, r.chrom_id = ident, CHROMS[chrom_id], chrom_id
    r.pos = rng.choice((rng.randrange(1, 5000), rng.randrange(1, 2_000_000_000), 4_000_000_000))
    r.mapq, r.score_mapq = rng.randrange(0, 255), rng.randrange(-5, 80)
    r.gamma = rng.choice((0.0, 1.0, 0.999999, 0.9999991, 0.5, 0.9, 1e-7, rng.random(), 1 - 10 ** -rng.uniform(0, 7)))
    r.mate, r.rev, r.duplicate = mate_no, rng.randrange(2), int(rng.random() < 0.2)
    r.cloud_id, r.cloud_bad, r.bc = rng.randrange(0, 100000), rng.randrange(2), bc
    r.read = bytes(rng.choice(b"ACGTN") for _ in range(l1)); r.qual = bytes(rng.choice(b"#,:FGH") for _ in range(l1)); r.read_len = l1
    r.mate_read = bytes(rng.choice(b"ACGTN") for _ in range(l2)); r.mate_qual = bytes(rng.choice(b"#,:FGH") for _ in range(l2)); r.mate_read_len = l2
    r.aln_pos, r.aln_rev, r.edit_dist = r.pos - 1 if rng.random() < 0.9 else rng.randrange(0, 10 ** 9), rng.randrange(2), rng.randrange(0, 30)
    r.n_cigar, r.cigar = len(ops), pool.cigar(ops)
    n_alts = rng.choice((0, 0, 0, 1, 3))
    if n_alts:
        alts = (sam.SamAlt * n_alts)()
        for a in alts:
            aops = rand_cigar(rng, l1)
            a.chrom, a.pos, a.edit_dist, a.rev, a.n_cigar, a.cigar = rng.choice(CHROMS), rng.randrange(1, 3_000_000_000), rng.randrange(0, 40), rng.randrange(2), len(aops), pool.cigar(aops)
        pool.keep.append(alts)
        r.alts, r.n_alts = alts, n_alts
    return r


def make_lines(rng, n_pairs, haplotag):
    pool = Pool()
    recs, lines = [], []
    for p in range(n_pairs):
        ident = b"@read%d/x" % p
        bc = (rng.randrange(1, 97) << 24 | rng.randrange(1, 97) << 16 | rng.randrange(1, 97) << 8 | rng.randrange(1, 97)) if haplotag \
            else rng.randrange(1 << 32)
        a, b = make_record(rng, pool, ident, bc, 0, haplotag), make_record(rng, pool, ident, bc, 1, haplotag)
        if rng.random() < 0.5:      # a proper-looking pair: same contig, opposite strands, close by
            b.chrom, b.chrom_id, b.rev = a.chrom, a.chrom_id, 1 - a.rev
            b.pos = max(1, a.pos + rng.randrange(-800, 800)) & 0xffffffff
            b.aln_pos = b.pos - 1
        recs += [a, b]
        kind = rng.randrange(4)      # the reference prints (best, mate) then (mate, best); either may be NULL
        pa, pb = C.pointer(a), C.pointer(b)
        null = C.POINTER(sam.SamRec)()
        if kind == 0:
            lines += [(pa, pb), (pb, pa)]
        elif kind == 1:
            lines += [(pa, null), (null, pa)]
        elif kind == 2:
            lines += [(null, pb), (pb, null)]
        else:
            lines += [(pb, pa), (pa, pb)]
    arr = (sam.SamLine * len(lines))()
    for i, (r, m) in enumerate(lines):
        arr[i].rec, arr[i].mate = r, m
    pool.keep += recs
    return arr, len(lines), pool


def oracle_text(arr, n, opts):
    L = O.lib()
    L.orc_sam_format.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    text, size = C.c_void_p(), C.c_size_t()
    assert L.orc_sam_format(C.cast(arr, C.c_void_p), n, C.cast(C.pointer(opts), C.c_void_p), C.byref(text), C.byref(size)) == 0
    out = C.string_at(text, size.value)
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(text)
    return out


@pytest.mark.parametrize("haplotag,rg,seed", [(False, None, 1), (False, b"grp1\tSM:x", 2), (True, b"hap", 3), (False, b"", 4)])
def test_formatter_equals_the_oracle(haplotag, rg, seed):
    rng = random.Random(seed)
    arr, n, pool = make_lines(rng, 400, haplotag)
    o = sam.default_opts()
    o.rg_id = rg
    if haplotag:
        o.is_haplotag, o.bc_len = 1, 12
    if seed == 2:
        o.bx_index = b"7"
    got, want = sam.format_lines(arr, n, o), oracle_text(arr, n, o)
    assert got == want
    assert got.count(b"\n") == n and b"\tXA:Z:" in got and b"\t=\t" in got


def test_large_batch_goes_through_the_threaded_path(tmp_path):
    rng = random.Random(9)
    arr, n, pool = make_lines(rng, 3000, False)      # 6000 lines: several threads, pieces laid end to end
    o = sam.default_opts()
    want = oracle_text(arr, n, o)
    assert sam.format_lines(arr, n, o) == want
    path = str(tmp_path / "out.sam")                 # and the same through ema_sam_write, after a header the caller wrote
    with open(path, "wb") as f:
        f.write(b"@HD\tVN:1.3\n")
        f.flush()
        assert sam.write_lines(f.fileno(), arr, n, o) == len(want)
    assert open(path, "rb").read() == b"@HD\tVN:1.3\n" + want


def test_known_line():
    """One hand-checked line, so that the two restatements cannot agree on a shared misreading of the easy parts."""
    pool = Pool()
    r = sam.SamRec()
    ops = [(5, 4), (95, 0)]
    r.ident, r.chrom, r.chrom_id, r.pos = b"q1", b"chr2", 1, 1000
    r.mapq, r.score_mapq, r.gamma = 60, 40, 0.99
    r.mate, r.rev, r.duplicate, r.cloud_id, r.cloud_bad, r.bc = 0, 1, 0, 7, 0, 0b11100100      # ACGT then A's
    r.read, r.qual, r.read_len = b"AACGN", b"12345", 5
    r.mate_read, r.mate_qual, r.mate_read_len = b"TT", b"##", 2
    r.aln_pos, r.aln_rev, r.edit_dist, r.n_cigar, r.cigar = 999, 1, 2, 2, pool.cigar(ops)
    arr = (sam.SamLine * 2)()
    arr[0].rec = C.pointer(r)
    arr[1].mate = C.pointer(r)
    o = sam.default_opts()
    o.bc_len = 6
    got = sam.format_lines(arr, 2, o)
    # MAPQ: (int)(-10 * log10(1 - 0.99)) is 19 in double arithmetic (1 - 0.99 = 0.01000000000000000888), below 40 and 60
    assert got == (b"q1\t89\tchr2\t1000\t19\t5S95M\t*\t0\t0\tNCGTT\t54321\tNM:i:2\tBX:Z:ACGTAA-1\tXG:f:0.99\tMI:i:7\tXF:i:0\n"
                   b"q1\t165\t*\t0\t0\t*\tchr2\t1000\t0\tTT\t##\tBX:Z:ACGTAA-1\n")


def test_bad_input_fails_loudly():
    pool = Pool()
    r = sam.SamRec()
    r.ident, r.chrom, r.read, r.qual, r.read_len, r.rev = b"q", b"c", b"AXG", b"###", 3, 1
    r.cigar, r.n_cigar = pool.cigar([(3, 0)]), 1
    arr = (sam.SamLine * 1)()
    arr[0].rec = C.pointer(r)
    with pytest.raises(RuntimeError, match="-7"):
        sam.format_lines(arr, 1, sam.default_opts())      # a base the reference's rc() asserts on
    empty = (sam.SamLine * 1)()
    with pytest.raises(RuntimeError, match="-1"):
        sam.format_lines(empty, 1, sam.default_opts())    # neither record nor mate


def test_header_equals_the_oracle():
    contigs = [(b"chr1", 248956422), (b"chrUn_KI270742v1", 186739), (b"c", 1)]
    argv = [b"ema", b"align", b"-r", b"ref.fa", b"-s", b"bucket 1/ema-bin-000"]
    L = O.lib()
    L.orc_sam_header.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_char_p, C.c_int,
                                 C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    for ctgs, rg in ((contigs, b"@RG\tID:grp1\tSM:x"), (contigs[:1], None), ([], None)):
        names = (C.c_char_p * max(1, len(ctgs)))(*[n for n, _ in ctgs])
        lens = (C.c_int32 * max(1, len(ctgs)))(*[l for _, l in ctgs])
        av = (C.c_char_p * len(argv))(*argv)
        text, size = C.c_void_p(), C.c_size_t()
        assert L.orc_sam_header(names, lens, len(ctgs), rg, b"0.6.2", len(argv), av, C.byref(text), C.byref(size)) == 0
        want = C.string_at(text, size.value)
        libc.free(text)
        got = sam.header(ctgs, rg, b"0.6.2", argv)
        assert got == want
        assert got.startswith(b"@HD\tVN:1.3\tSO:unsorted\n") and got.endswith(b"CL:ema align -r ref.fa -s bucket 1/ema-bin-000\n")
        assert got.count(b"@SQ") == len(ctgs) and (b"@RG" in got) == (rg is not None)
