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
