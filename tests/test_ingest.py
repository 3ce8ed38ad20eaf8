"""Bucket reader (include/ema_ingest.h): the product's parallel reader against the oracle's
line-by-line restatement of read_special_fastq (reference src/align.c:751-843) on the same files, the oracle's util.c
restatements against the reference's own util.c where it was compiled (oracle/Makefile `ref`, built outside the repository), and the error behaviour where the
reference has undefined behaviour.  CPU only: the reader is host code."""
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from ema_amd import ingest, synth

BASES = b"ACGT"


def rand_seq(rng, n, alphabet=BASES):
    return bytes(rng.choice(alphabet) for _ in range(n))


def make_bucket(rng, n_barcodes, bc_len=16, haplotag=False, mixed_case=False, seps=b" ", newline=b"\n", tail_newline=True,
                max_len=150, extra_field=False):
    """Lines of a special FASTQ, shuffled; barcodes repeat so that equal keys exercise the stable order."""
    lines = []
    for _ in range(n_barcodes):
        if haplotag:
            bc = b"A%02dC%02dB%02dD%02d" % tuple(rng.randrange(1, 97) for _ in range(4))
        else:
            bc = rand_seq(rng, bc_len)
        for _ in range(rng.randrange(1, 6)):
            b = bc
            if mixed_case and rng.random() < 0.3:
                b = bytes(c + 32 if rng.random() < 0.5 else c for c in bc)      # some bases in lower case
            l1, l2 = rng.randrange(1, max_len + 1), rng.randrange(1, max_len + 1)
            ident = b"@s%d/%d" % (len(lines), rng.randrange(10 ** rng.randrange(1, 9)))
            f = [b, ident, rand_seq(rng, l1, b"ACGTN"), rand_seq(rng, l1, b"FGH#,:"), rand_seq(rng, l2, b"ACGTN"), rand_seq(rng, l2, b"FGH#,:")]
            if extra_field:
                f.append(b"ignored tail")
            line = b""
            for k, x in enumerate(f):
                line += x + (bytes([rng.choice(seps)]) if k + 1 < len(f) else b"")
            lines.append(line)
    rng.shuffle(lines)
    text = newline.join(lines) + (newline if tail_newline else b"")
    return text


def check_same(path, text, bc_len=16, haplotag=False):
    with open(path, "wb") as f:
        f.write(text)
    want, groups = O.read_special_fastq(path, bc_len, haplotag)
    for got in (ingest.read_bucket(path, bc_len, haplotag), ingest.parse_bucket(text, bc_len, haplotag)):
        assert got.n_pairs == len(want)
        for i, (bc, ident, r1, q1, r2, q2) in enumerate(want):
            assert int(got.bc[i]) == bc and got.ident(i) == ident, i
            assert got.read(2 * i) == r1 and got.qual(2 * i) == q1 and got.read(2 * i + 1) == r2 and got.qual(2 * i + 1) == q2, i
        assert [(int(a), int(b - a)) for a, b in zip(got.group_off[:-1], got.group_off[1:])] == groups
    return want, groups


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_reader_equals_the_oracle_on_10x_buckets(tmp_path, seed):
    rng = random.Random(seed)
    want, groups = check_same(str(tmp_path / "b.fq"), make_bucket(rng, 300))
    assert len(groups) > 250 and any(n > 1 for _, n in groups)
    assert [w[0] for w in want] != sorted(w[0] for w in want)      # the order is the text order of barcodes, not of their codes


def test_reader_on_format_variants(tmp_path):
    rng = random.Random(7)
    p = str(tmp_path / "b.fq")
    check_same(p, make_bucket(rng, 120, seps=b" \t"))                        # any whitespace separates (isspace)
    check_same(p, make_bucket(rng, 120, newline=b"\r\n"))                     # '\r' ends the last field
    check_same(p, make_bucket(rng, 120, tail_newline=False))                  # last line without a newline
    check_same(p, make_bucket(rng, 120, extra_field=True))                    # anything after the sixth field is ignored
    check_same(p, make_bucket(rng, 120, mixed_case=True))                     # 'a' encodes as 'A' but sorts after 'T'
    check_same(p, make_bucket(rng, 120, max_len=255))                         # up to the engine's read length
    check_same(p, make_bucket(rng, 200, bc_len=12, haplotag=True), 12, True)  # haplotag codes
    check_same(p, make_bucket(rng, 200, bc_len=20), 20)                       # dbs
    check_same(p, make_bucket(rng, 200, bc_len=18), 18)                       # tellseq
    check_same(p, make_bucket(rng, 40, bc_len=3), 3)                          # many equal keys: file order within a barcode
    check_same(p, b"")                                                        # empty bucket
    check_same(p, make_bucket(rng, 1))


def test_mixed_case_barcodes_split_groups_as_in_the_reference(tmp_path):
    """Lower-case barcodes sort apart from their upper-case twins (strncmp on the text) but encode equal: the
    reference's grouping walks runs of equal code, so the two runs are two groups unless they happen to be adjacent."""
    lines = [b"ACGTACGTACGTACGT r1 AC FF GT FF", b"acgtacgtacgtacgt r2 AC FF GT FF", b"CCGTACGTACGTACGT r3 AC FF GT FF",
             b"ACGTACGTACGTACGT r4 A F G F"]
    want, groups = check_same(str(tmp_path / "b.fq"), b"\n".join(lines) + b"\n")
    assert [w[1] for w in want] == [b"r1", b"r4", b"r3", b"r2"]
    assert groups == [(0, 2), (2, 1), (3, 1)] and want[0][0] == want[3][0]


def test_large_bucket_goes_through_the_threaded_passes(tmp_path):
    rng = random.Random(11)
    text = make_bucket(rng, 30000, max_len=20)      # ~90 K lines: several sort chunks and a merge tree
    want, groups = check_same(str(tmp_path / "b.fq"), text)
    assert len(want) > 60000


@pytest.mark.parametrize("bad,what", [
    (b"ACGTACGTACGTACGT id AC FF GT\n", "quality"),                      # five fields: the sixth reads as empty
    (b"ACGTACGTACGTACGT id AC FF\n", "fewer than six"),
    (b"\n", "fewer than six"),
    (b"ACGTACGTACGTACG id AC FF GT FF\n", "bc_len"),
    (b"ACGTACGTACGTACGN id AC FF GT FF\n", "ACGT"),
    (b"ACGTACGTACGTACGT  AC FF GT FF\n", "empty identifier"),
    (b"ACGTACGTACGTACGT " + b"i" * 150 + b" AC FF GT FF\n", "149"),
    (b"ACGTACGTACGTACGT id " + b"A" * 256 + b" " + b"F" * 256 + b" GT FF\n", "max_read_len"),
    (b"ACGTACGTACGTACGT id AC F GT FF\n", "quality"),
    (b"ACGTACGTACGTACGT id AC FF GT FF " + b"x" * 5000 + b"\n", "5000"),
])
def test_malformed_lines_fail_loudly(bad, what):
    good = b"ACGTACGTACGTACGA ok AC FF GT FF\n"
    with pytest.raises(ingest.BucketError) as e:
        ingest.parse_bucket(good + bad + good)
    assert e.value.code == ingest.EMA_EFORMAT and "line 2" in str(e.value) and what in str(e.value)


def test_missing_file_is_an_io_error(tmp_path):
    with pytest.raises(ingest.BucketError) as e:
        ingest.read_bucket(str(tmp_path / "nope.fq"))
    assert e.value.code == ingest.EMA_EIO


def test_barcode_codes_round_trip():
    rng = random.Random(5)
    for n in (1, 12, 16, 18, 20, 32):
        for _ in range(50):
            bc = rand_seq(rng, n)
            v = ingest.encode_barcode(bc)
            assert v == O.oracle_encode_bc(bc) and ingest.decode_barcode(v, n) == bc == O.oracle_decode_bc(v, n)
    for _ in range(50):
        bc = b"A%02dC%02dB%02dD%02d" % tuple(rng.randrange(0, 100) for _ in range(4))
        v = ingest.encode_barcode(bc, True)
        assert v == O.oracle_encode_bc(bc, True) and ingest.decode_barcode(v, 12, True) == bc == O.oracle_decode_bc(v, 12, True)
    assert ingest.encode_barcode(b"A" * 16) == 0      # the all-A barcode is the reference's sentinel code (src/align.c:1060)


@pytest.mark.skipif(not os.path.exists(O.REF_UTIL), reason="$TMPDIR/ema_ref not built (needs /root/reference at build time)")
def test_oracle_util_restatements_equal_the_reference_util_c():
    """Pins oracle/ingest.c's copy_until_space / encode_bc / decode_bc to the reference's own compiled src/util.c."""
    ref = O.RefUtil()
    rng = random.Random(9)
    for n in (12, 16, 18, 20):
        for _ in range(200):
            bc = bytes(c + 32 if rng.random() < 0.2 else c for c in rand_seq(rng, n))
            v = ref.encode_bc(bc)
            assert v == O.oracle_encode_bc(bc) == ingest.encode_barcode(bc)
            assert ref.decode_bc(v, n) == O.oracle_decode_bc(v, n) == ingest.decode_barcode(v, n) == bc.upper()
    for _ in range(200):
        bc = b"A%02dC%02dB%02dD%02d" % tuple(rng.randrange(0, 100) for _ in range(4))
        v = ref.encode_bc(bc, True)
        assert v == O.oracle_encode_bc(bc, True) == ingest.encode_barcode(bc, True)
        assert ref.decode_bc(v, 12, True) == O.oracle_decode_bc(v, 12, True) == ingest.decode_barcode(v, 12, True)
    for line in (b"a b c d e f\n", b"a\tb  c\rd\n", b"abc", b" x y", b"one two\n", b"BC id READ QUAL READ2 QUAL2 rest of line\n"):
        for k in (1, 2, 3):      # stay within the fields the line has: beyond them the reference reads past the terminator
            assert ref.copy_until_space(line, k) == O.oracle_copy_until_space(line, k), (line, k)


def test_mutated_buckets_never_crash_the_reader():
    """Random damage to a valid bucket: the reader either fails with EMA_EFORMAT naming a line, or returns a bucket whose
    arrays are consistent (the oracle is not consulted here: on malformed input the reference, and so its restatement,
    has undefined behaviour)."""
    rng = random.Random(13)
    base = make_bucket(rng, 60)
    n_ok = n_bad = 0
    for trial in range(300):
        b = bytearray(base)
        for _ in range(rng.randrange(1, 6)):
            kind, at = rng.randrange(7), rng.randrange(len(b))
            if kind >= 5:      # harmless more often than not: a base or quality character for whatever was there
                b[at] = rng.choice(b"ACGTF")
            elif kind == 0:
                b[at] = rng.choice(b" \t\n\r\0AXn#")
            elif kind == 1:
                del b[at:at + rng.randrange(1, 40)]
            elif kind == 2:
                b[at:at] = bytes(rng.choice(b"ACGT \n") for _ in range(rng.randrange(1, 30)))
            elif kind == 3:
                b[at:at] = b"A" * rng.choice((200, 300, 6000))
            else:
                b = b[:at]
        try:
            got = ingest.parse_bucket(bytes(b))
        except ingest.BucketError as e:
            assert e.code == ingest.EMA_EFORMAT and "line " in str(e)
            n_bad += 1
            continue
        n_ok += 1
        n = got.n_pairs
        assert len(got.off) == 2 * n + 1 and len(got.id_off) == n + 1 and got.off[0] == 0 and got.id_off[0] == 0
        assert (np.diff(got.off.astype(np.int64)) >= 0).all() and (np.diff(got.id_off.astype(np.int64)) > 0).all()
        assert len(got.bases) == len(got.quals) == got.off[-1] and len(got.ids) == got.id_off[-1]
        assert got.group_off[0] == 0 and got.group_off[-1] == n and (np.diff(got.group_off.astype(np.int64)) > 0).all()
        assert (np.diff(got.off.astype(np.int64)) <= 255).all()
    assert n_ok >= 5 and n_bad > 100


def test_the_first_bad_line_is_the_one_reported():
    good = b"ACGTACGTACGTACGA ok AC FF GT FF\n"
    text = good + b"ACGTACGTACGTACGN id AC FF GT FF\n" + good + b"ACGTACGTACGTACGT id AC FF\n"
    with pytest.raises(ingest.BucketError) as e:
        ingest.parse_bucket(text)
    assert "line 2" in str(e.value) and "ACGT" in str(e.value)


def test_fixed_width_bucket_writer_reads_back_like_the_per_pair_writer(tmp_path):
    """synth.write_special_fastq_fixed (one array write; identifiers zero-padded) and synth.write_special_fastq (a line per pair)
    produce buckets that differ in nothing but the identifiers' spelling: what bench.py's bucket-files-to-SAM leg feeds."""
    ctg = synth.make_genome([120000], seed=5)
    pairs = synth.make_pairs(ctg, 700, seed=9)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    synth.write_special_fastq(a, pairs)
    synth.write_special_fastq_fixed(b, pairs)
    ba, bb = ingest.read_bucket(a), ingest.read_bucket(b)
    assert ba.n_pairs == bb.n_pairs == 700
    assert np.array_equal(ba.bc, bb.bc) and np.array_equal(ba.off, bb.off)
    # equal barcodes keep file order, and both files list the pairs in the same order: the payloads line up
    assert bytes(ba.bases) == bytes(bb.bases) and bytes(ba.quals) == bytes(bb.quals)
    assert [int(ba.ident(i)[2:]) for i in range(700)] == [int(bb.ident(i)[2:]) for i in range(700)]


def _fq(name, read, qual=None):
    return b"@" + name + b"\n" + read + b"\n+\n" + (qual if qual is not None else b"F" * len(read)) + b"\n"


def test_fastq_reader_of_align_1_2(tmp_path):
    """ema_fastq_read (`ema align -1 [-2]`, reference src/align.c:637-744, src/techs.c:5-69): barcode after the last ':' of the name,
    identifier cut at that ':' and at the first blank, groups = runs of equal barcode in FILE order, both mates agreeing.  (The
    whole path on such input is held to the reference's own output in tests/test_golden_sam.py.)"""
    bc1, bc2 = b"ACGTACGTACGTACGT", b"TTTTCCCCGGGGAAAA"
    recs = [(b"s1", bc1, b"ACGTACGTAC", b"GGGGTTTTAA"), (b"s2", bc1, b"AAAACCCC", b"CCCCAAAATT"), (b"s3", bc2, b"ACGT", b"TTGA"), (b"s4", bc1, b"GGGG", b"CCCC")]
    f1 = b"".join(_fq(n + b" 1:N:0:" + bc, r1) for n, bc, r1, _r2 in recs)
    f2 = b"".join(_fq(n + b" 2:N:0:" + bc, r2) for n, bc, _r1, r2 in recs)
    p1, p2 = tmp_path / "a.fq", tmp_path / "b.fq"
    p1.write_bytes(f1); p2.write_bytes(f2)
    b = ingest.read_fastq(str(p1), str(p2))
    assert b.n_pairs == 4 and b.group_off.tolist() == [0, 2, 3, 4]      # the third barcode run is bc1 again: file order, not sorted
    assert [b.ident(i) for i in range(4)] == [b"@s1", b"@s2", b"@s3", b"@s4"]
    assert b.read(0) == b"ACGTACGTAC" and b.read(1) == b"GGGGTTTTAA" and b.read(6) == b"GGGG"
    assert b.bc[0] == ingest.encode_barcode(bc1) and b.bc[2] == ingest.encode_barcode(bc2)
    # the same pairs interleaved, names in the plain form name:BARCODE, CRLF line ends, no final newline
    inter = b"".join(_fq(n + b":" + bc, r1) + _fq(n + b":" + bc, r2) for n, bc, r1, r2 in recs).replace(b"\n", b"\r\n")[:-2]
    p3 = tmp_path / "ab.fq"
    p3.write_bytes(inter)
    c = ingest.read_fastq(str(p3))
    assert c.n_pairs == 4 and c.bases.tobytes() == b.bases.tobytes() and c.ids.tobytes() == b.ids.tobytes() and c.bc.tolist() == b.bc.tolist()
    # where the reference asserts or overruns a buffer, the reader names the record
    for bad, what in ((f1[:-3], "differ in length"), (f1[:-6], "truncated"), (f1.replace(b"@s3", b"s3"), "does not start"), (f1 + b"@x:ACGT\nAC\n+\nFF\n", "shorter"),
                      (f1.replace(bc2, b"TTTTCCCCGGGGAAAN"), "outside ACGT")):
        p1.write_bytes(bad)
        with pytest.raises(ingest.BucketError) as e:
            ingest.read_fastq(str(p1), str(p2))
        assert what in str(e.value), str(e.value)
    p1.write_bytes(f1)
    p2.write_bytes(f2.replace(b"s2 2:N:0:" + bc1, b"s2 2:N:0:" + bc2))
    with pytest.raises(ingest.BucketError, match="different barcodes"):
        ingest.read_fastq(str(p1), str(p2))
    p2.write_bytes(f2[:len(f2) // 2 + 3])
    with pytest.raises(ingest.BucketError):
        ingest.read_fastq(str(p1), str(p2))
    p3.write_bytes(_fq(b"s1:" + bc1, b"ACGT") * 3)
    with pytest.raises(ingest.BucketError, match="odd number"):
        ingest.read_fastq(str(p3))


def test_fastq_reader_integer_barcodes_of_tru_and_cpt(tmp_path):
    """ema_fastq_read name styles 2 and 3 (`-p tru`, `-p cpt`; reference src/techs.c:56-68): extract_bc_truseq takes atoi() of the name
    behind its '@' and leaves the name whole; extract_bc_cptseq cuts the name at its last ':' and takes atoi() of what follows that ':'
    and two more characters.  (The whole path on such input: tests/golden/sam/tru_fastq_many_clouds, cpt_fastq_density_opt.)"""
    recs = [(b"12_a", b"ACGTAC", b"GGTTAA"), (b"12_b", b"AAAACC", b"CCAATT"), (b"7x", b"ACGT", b"TTGA"), (b"+7y", b"GGGG", b"CCCC"), (b"x9", b"AC", b"GT"), (b"-3z", b"AC", b"GT")]
    p = tmp_path / "t.fq"
    p.write_bytes(b"".join(_fq(n, r1) + _fq(n, r2) for n, r1, r2 in recs))
    b = ingest.read_fastq(str(p), bc_len=0, name_style=2)
    assert b.n_pairs == 6 and b.group_off.tolist() == [0, 2, 4, 5, 6]      # 12 12 | 7 7 | 0 (no number) | -3
    assert b.bc.tolist() == [12, 12, 7, 7, 0, (1 << 64) - 3]               # bc_t = uint64_t of atoi()'s int
    assert [b.ident(i) for i in range(6)] == [b"@" + n for n, _r1, _r2 in recs]
    recs = [(b"r1:BC41", b"ACGTAC", b"GGTTAA"), (b"r2:xy41tail", b"AAAACC", b"CCAATT"), (b"a:b:BC5", b"ACGT", b"TTGA"), (b"q:B", b"AC", b"GT")]
    p.write_bytes(b"".join(_fq(n, r1) + _fq(n, r2) for n, r1, r2 in recs))
    c = ingest.read_fastq(str(p), bc_len=0, name_style=3)
    assert c.bc.tolist() == [41, 41, 5, 0] and c.group_off.tolist() == [0, 2, 3, 4]
    assert [c.ident(i) for i in range(4)] == [b"@r1", b"@r2", b"@a:b", b"@q"]
    p.write_bytes(_fq(b"nocolon", b"AC") * 2)
    with pytest.raises(ingest.BucketError, match="no ':'"):
        ingest.read_fastq(str(p), bc_len=0, name_style=3)
    with pytest.raises(ingest.BucketError):
        ingest.read_fastq(str(p), bc_len=0, name_style=0)      # the ACGT platforms need their barcode length
