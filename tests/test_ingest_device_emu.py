"""The device reader's kernels (csrc/ingest_kernels.hpp: newline count, field scan with the reader's checks and the barcode's sort
code, lengths in sorted order, gather + barcode encoding) under the host SIMT interpreter -- the driver's library passes (select,
radix sort, prefix sums) replaced by plain loops in tests/emu/harness.cpp -- against the host reader on tests/test_ingest.py's
buckets: array for array where the kernels take the bucket, flagged irregular exactly where the host reader refuses the file or the
driver hands it over (NUL bytes).  CPU only; the GPU runs the real driver in tests/test_gpu_ingest_device.py.  The interpreter is
test infrastructure, not parity evidence."""
import ctypes as C
import random

import numpy as np
import pytest

import emu_lib
from ema_amd import ingest
from test_ingest import make_bucket


def emu_read(text, bc_len=16, max_read_len=255, haplotag=False):
    L = emu_lib.lib()
    L.emu_ingest.restype = C.c_int
    L.emu_ingest.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6 + [C.POINTER(C.c_uint32)]
    cap = text.count(b"\n") + 2
    bc, off, id_off = np.zeros(cap, np.uint64), np.zeros(2 * cap + 1, np.uint32), np.zeros(cap + 1, np.uint32)
    bases, quals, ids = (np.zeros(len(text) + 8, np.uint8) for _ in range(3))
    n = C.c_uint32()
    rc = L.emu_ingest(text, len(text), bc_len, int(haplotag), max_read_len, cap, bc.ctypes.data, off.ctypes.data, id_off.ctypes.data, bases.ctypes.data,
                      quals.ctypes.data, ids.ctypes.data, C.byref(n))
    n = n.value
    return rc, dict(bc=bc[:n], off=off[:2 * n + 1], id_off=id_off[:n + 1], bases=bases[:off[2 * n]], quals=quals[:off[2 * n]], ids=ids[:id_off[n]])


def same(text, bc_len=16, haplotag=False):
    want = ingest.parse_bucket(text, bc_len, haplotag)
    rc, got = emu_read(text, bc_len, haplotag=haplotag)
    assert rc == 0
    for name, arr in got.items():
        assert np.array_equal(arr, getattr(want, name)), name


def test_interpreted_kernels_equal_the_host_reader():
    for seed in (1, 2):
        same(make_bucket(random.Random(seed), 200))
    rng = random.Random(7)
    same(make_bucket(rng, 80, seps=b" \t"))
    same(make_bucket(rng, 80, newline=b"\r\n"))
    same(make_bucket(rng, 80, tail_newline=False))
    same(make_bucket(rng, 80, extra_field=True))
    same(make_bucket(rng, 80, mixed_case=True))
    same(make_bucket(rng, 80, max_len=255))
    same(make_bucket(rng, 100, bc_len=20), 20)
    same(make_bucket(rng, 40, bc_len=3), 3)
    same(make_bucket(rng, 1))
    same(make_bucket(rng, 150, bc_len=12, haplotag=True), 12, True)      # the twelve bytes as the key: two stable sorts on the device
    same(make_bucket(random.Random(3), 60, bc_len=12, haplotag=True).replace(b"A0", b"Ax"), 12, True)      # ... whatever the bytes are (the reference's macros)


@pytest.mark.parametrize("bad", [
    b"ACGTACGTACGTACGT id AC FF GT\n", b"ACGTACGTACGTACGT id AC FF\n", b"\n", b"ACGTACGTACGTACG id AC FF GT FF\n",
    b"ACGTACGTACGTACGN id AC FF GT FF\n", b"ACGTACGTACGTACGT  AC FF GT FF\n", b"ACGTACGTACGTACGT " + b"i" * 150 + b" AC FF GT FF\n",
    b"ACGTACGTACGTACGT id " + b"A" * 256 + b" " + b"F" * 256 + b" GT FF\n", b"ACGTACGTACGTACGT id AC F GT FF\n",
    b"ACGTACGTACGTACGT id AC FF GT FF " + b"x" * 5000 + b"\n", b"ACGTACGTACGTACGT id AC FF GT FF x\0y\n"])
def test_what_the_host_reader_refuses_is_flagged(bad):
    good = b"ACGTACGTACGTACGA ok AC FF GT FF\n"
    rc, _ = emu_read(good + bad + good)
    assert rc > 0      # bit 0: a NUL byte; bit 1: a line the checks refuse -- the driver then calls ema_bucket_read
    rc, _ = emu_read(good + good)
    assert rc == 0
