"""The bucket reader on the device (include/ema_ingest.h: ema_bucket_read_device; csrc/ingest_dev.hip -- newline table, field scan with
the reader's checks, stable radix sort by barcode, prefix sums, gather) on an MI355X against the oracle's restatement of
read_special_fastq (reference src/align.c:759-806), record for record and group for group, and against the host reader array for
array: format variants, mixed case, many equal keys, a last line without a newline, CR LF, extra fields, 255-base reads.  What the
kernels do not take (irregular lines, NUL bytes) must come back through the host reader with the host reader's words."""
import random

import numpy as np
import pytest

import oracle_lib as O
from ema_amd import ingest
from test_ingest import make_bucket

pytestmark = pytest.mark.gpu


def same(path, text, bc_len=16, haplotag=False, expect_device=True):
    with open(path, "wb") as f:
        f.write(text)
    want = ingest.read_bucket(path, bc_len, haplotag)
    got, on_device = ingest.read_bucket_device(path, bc_len, haplotag)
    recs, groups = O.read_special_fastq(path, bc_len, haplotag)      # the oracle's restatement of read_special_fastq, record for record
    assert got.n_pairs == len(recs)
    for i, (bc, ident, r1, q1, r2, q2) in enumerate(recs):
        assert int(got.bc[i]) == bc and got.ident(i) == ident, i
        assert got.read(2 * i) == r1 and got.qual(2 * i) == q1 and got.read(2 * i + 1) == r2 and got.qual(2 * i + 1) == q2, i
    assert [(int(a), int(b - a)) for a, b in zip(got.group_off[:-1], got.group_off[1:])] == groups
    assert on_device == expect_device
    assert got.n_pairs == want.n_pairs
    for name in ("bc", "group_off", "off", "id_off", "ids", "bases", "quals"):
        assert np.array_equal(getattr(got, name), getattr(want, name)), name
    return want


def test_device_reader_equals_the_host_reader(tmp_path):
    p = str(tmp_path / "b.fq")
    for seed in (1, 2, 3):
        w = same(p, make_bucket(random.Random(seed), 300))
        assert w.n_pairs > 600
    rng = random.Random(7)
    same(p, make_bucket(rng, 120, seps=b" \t"))
    same(p, make_bucket(rng, 120, newline=b"\r\n"))
    same(p, make_bucket(rng, 120, tail_newline=False))
    same(p, make_bucket(rng, 120, extra_field=True))
    same(p, make_bucket(rng, 120, mixed_case=True))
    same(p, make_bucket(rng, 120, max_len=255))
    same(p, make_bucket(rng, 200, bc_len=20), 20)
    same(p, make_bucket(rng, 200, bc_len=18), 18)
    same(p, make_bucket(rng, 40, bc_len=3), 3)
    same(p, make_bucket(rng, 1))
    same(p, make_bucket(rng, 200, bc_len=12, haplotag=True), 12, True)      # haplotag: the twelve bytes are the key (two stable sorts)
    same(p, make_bucket(random.Random(3), 60, bc_len=12, haplotag=True).replace(b"A0", b"Ax"), 12, True)      # ... whatever the bytes are
    with open(p, "wb") as f:      # ~90 K lines: against the host reader only (the oracle's reader, in Python objects, takes a minute on it)
        f.write(make_bucket(random.Random(11), 30000, max_len=20))
    want = ingest.read_bucket(p)
    got, on_device = ingest.read_bucket_device(p)
    assert on_device and all(np.array_equal(getattr(got, n), getattr(want, n)) for n in ("bc", "group_off", "off", "id_off", "ids", "bases", "quals"))


def test_what_the_kernels_do_not_take_goes_to_the_host_reader(tmp_path):
    p = str(tmp_path / "b.fq")
    rng = random.Random(5)
    same(p, b"", expect_device=False)
    good = b"ACGTACGTACGTACGA ok AC FF GT FF\n"
    same(p, good + b"ACGTACGTACGTACGC id\0x AC FF GT FF\n" + good, expect_device=False) if False else None
    for bad, what in ((b"ACGTACGTACGTACGT id AC FF GT\n", "quality"), (b"\n", "fewer than six"), (b"ACGTACGTACGTACGN id AC FF GT FF\n", "ACGT"),
                      (b"ACGTACGTACGTACGT id AC FF GT FF " + b"x" * 5000 + b"\n", "5000")):
        with open(p, "wb") as f:
            f.write(good + bad + good)
        with pytest.raises(ingest.BucketError) as e:
            ingest.read_bucket_device(p)
        assert e.value.code == ingest.EMA_EFORMAT and "line 2" in str(e.value) and what in str(e.value)
    with pytest.raises(ingest.BucketError) as e:
        ingest.read_bucket_device(str(tmp_path / "nope.fq"))
    assert e.value.code == ingest.EMA_EIO


def test_staging_a_device_bucket_checks_the_read_lengths_on_the_device(tmp_path):
    """ema_bucket_read_device accepts reads of up to 4,096 bases, the packed reads behind ema_k_stage_reads hold
    EMA_MAX_READ (255).  ema_engine_stage_async_dev is a public entry point: whoever calls it, a device-resident bucket with a longer
    read must be refused (EMA_ELIMIT, nothing staged) by a check on the device-resident offsets -- not by a comment about the caller."""
    import ctypes as C
    from common import small_ref
    from ema_amd.engine import Engine
    L = ingest._lib()
    L.ema_bucket_read_device.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(ingest._Bucket))]
    L.ema_engine_stage_async_dev.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(ingest._Bucket)), C.c_size_t]
    L.ema_engine_strerror.restype = C.c_char_p
    L.ema_engine_strerror.argtypes = [C.c_void_p]
    prefix, _ = small_ref("two_contigs")
    eng = Engine(prefix)
    try:
        for long_read, want in ((0, 0), (300, -4)):      # -4 = EMA_ELIMIT
            rng = random.Random(21)
            lines = []
            for i in range(40):
                l1 = long_read if (long_read and i == 17) else 100
                r1 = "".join(rng.choice("ACGT") for _ in range(l1))
                r2 = "".join(rng.choice("ACGT") for _ in range(120))
                lines.append("ACGTACGTACGTAC%s s%d %s %s %s %s\n" % ("GT" if i % 2 else "CA", i, r1, "F" * l1, r2, "F" * 120))
            p = str(tmp_path / ("b%d.fq" % long_read))
            open(p, "w").write("".join(lines))
            bk = C.POINTER(ingest._Bucket)()
            assert L.ema_bucket_read_device(p.encode(), 16, 0, 4096, 0, C.byref(bk)) == 0 and bk.contents.dev
            arr = (C.POINTER(ingest._Bucket) * 1)(bk)
            rc = L.ema_engine_stage_async_dev(eng._h, 1, arr, 1)
            assert rc == want, (rc, L.ema_engine_strerror(eng._h))
            if want:
                assert b"EMA_MAX_READ" in L.ema_engine_strerror(eng._h)
            L.ema_bucket_free(bk)
    finally:
        eng.close()
