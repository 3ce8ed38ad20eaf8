"""Host mirror of the bucket reader (include/ema_ingest.h): the reference's read_special_fastq + barcode grouping
(reference src/align.c:759-843) as one call whose result is the engine's input layout.  ctypes over the C ABI in
libema_engine.so; no fallback -- a missing library raises."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import engine as _engine

EMA_EIO, EMA_EFORMAT = -6, -7


class _Bucket(C.Structure):
    _fields_ = [("n_pairs", C.c_size_t), ("n_groups", C.c_size_t), ("group_off", C.POINTER(C.c_uint64)),
                ("bc", C.POINTER(C.c_uint64)), ("off", C.POINTER(C.c_uint32)), ("bases", C.POINTER(C.c_char)),
                ("quals", C.POINTER(C.c_char)), ("id_off", C.POINTER(C.c_uint32)), ("ids", C.POINTER(C.c_char)), ("dev", C.c_void_p)]


@dataclass
class Bucket:
    """One barcode bucket, ordered as the reference orders it.  bases/off are what Engine.align_pairs takes."""
    bc: np.ndarray          # u64 per pair
    group_off: np.ndarray   # u64, n_groups + 1
    off: np.ndarray         # u32, 2 * n_pairs + 1
    bases: np.ndarray       # u8
    quals: np.ndarray       # u8, same offsets as bases
    id_off: np.ndarray      # u32, n_pairs + 1
    ids: np.ndarray         # u8

    @property
    def n_pairs(self) -> int:
        return len(self.bc)

    def read(self, r: int) -> bytes:
        return self.bases[self.off[r]:self.off[r + 1]].tobytes()

    def qual(self, r: int) -> bytes:
        return self.quals[self.off[r]:self.off[r + 1]].tobytes()

    def ident(self, p: int) -> bytes:
        return self.ids[self.id_off[p]:self.id_off[p + 1]].tobytes()


def _lib():
    L = _engine.load_library()
    if not getattr(L, "_ingest_bound", False):
        L.ema_bucket_read.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(_Bucket))]
        L.ema_bucket_parse.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(_Bucket))]
        L.ema_bucket_free.argtypes = [C.POINTER(_Bucket)]
        L.ema_bucket_last_error.restype = C.c_char_p
        L.ema_barcode_encode.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
        L.ema_barcode_decode.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_char_p]
        L._ingest_bound = True
    return L


class BucketError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"bucket reader: {msg} (code {code})")
        self.code = code


def _take(L, p) -> Bucket:
    try:
        b = p.contents
        n, g = b.n_pairs, b.n_groups

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
        off = arr(b.off, 2 * n + 1, np.uint32)
        id_off = arr(b.id_off, n + 1, np.uint32)
        return Bucket(arr(b.bc, n, np.uint64), arr(b.group_off, g + 1, np.uint64), off, arr(b.bases, int(off[-1]), np.uint8),
                      arr(b.quals, int(off[-1]), np.uint8), id_off, arr(b.ids, int(id_off[-1]), np.uint8))
    finally:
        L.ema_bucket_free(p)


def read_bucket(path: str, bc_len: int = 16, is_haplotag: bool = False, max_read_len: int = 255) -> Bucket:
    L = _lib()
    p = C.POINTER(_Bucket)()
    rc = L.ema_bucket_read(path.encode(), bc_len, int(is_haplotag), max_read_len, C.byref(p))
    if rc != 0:
        raise BucketError(rc, L.ema_bucket_last_error().decode())
    return _take(L, p)


def read_bucket_device(path: str, bc_len: int = 16, is_haplotag: bool = False, max_read_len: int = 255, device: int = 0):
    """ema_bucket_read_device: the reader's parse, sort and gather as kernels (csrc/ingest_dev.hip).  Returns (Bucket, on_device): the
    bucket with its reads and qualities copied back from the device (ema_bucket_dev_fetch) so that it compares with read_bucket's, and
    whether the kernels took it (False: the library handed the file to the host reader -- haplotag, long barcodes, an irregular line)."""
    L = _lib()
    L.ema_bucket_read_device.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(_Bucket))]
    L.ema_bucket_dev_fetch.argtypes = [C.POINTER(_Bucket), C.c_void_p, C.c_void_p]
    L.ema_bucket_dev_last_error.restype = C.c_char_p
    p = C.POINTER(_Bucket)()
    rc = L.ema_bucket_read_device(path.encode(), bc_len, int(is_haplotag), max_read_len, device, C.byref(p))
    if rc != 0:
        raise BucketError(rc, (L.ema_bucket_last_error() or L.ema_bucket_dev_last_error()).decode())
    b = p.contents
    on_device = bool(b.dev)
    if not on_device:
        return _take(L, p), False
    assert not b.bases and not b.quals
    nb = int(b.off[2 * b.n_pairs])
    bases, quals = np.zeros(nb + 1, np.uint8), np.zeros(nb + 1, np.uint8)
    rc = L.ema_bucket_dev_fetch(p, bases.ctypes.data, quals.ctypes.data)
    if rc != 0:
        L.ema_bucket_free(p)
        raise BucketError(rc, "ema_bucket_dev_fetch failed")
    b.bases, b.quals = C.cast(bases.ctypes.data, C.POINTER(C.c_char)), C.cast(quals.ctypes.data, C.POINTER(C.c_char))
    try:
        n, g = b.n_pairs, b.n_groups

        def arr(ptr, count, dt):
            if count == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(count * np.dtype(dt).itemsize,)).view(dt).copy()
        off = arr(b.off, 2 * n + 1, np.uint32)
        id_off = arr(b.id_off, n + 1, np.uint32)
        out = Bucket(arr(b.bc, n, np.uint64), arr(b.group_off, g + 1, np.uint64), off, bases[:nb].copy(), quals[:nb].copy(), id_off,
                     arr(b.ids, int(id_off[-1]), np.uint8))
    finally:
        b.bases, b.quals = None, None      # (numpy's memory: not the library's to free)
        L.ema_bucket_free(p)
    return out, True


def read_fastq(path1: str, path2: str | None = None, bc_len: int = 16, is_haplotag: bool = False, max_read_len: int = 255,
               name_style: int = 0) -> Bucket:
    """ema_fastq_read: barcode-sorted FASTQ as `ema align -1 [-2]` takes it (path2 None: interleaved)."""
    L = _lib()
    L.ema_fastq_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(_Bucket))]
    p = C.POINTER(_Bucket)()
    rc = L.ema_fastq_read(path1.encode(), path2.encode() if path2 else None, name_style, bc_len, int(is_haplotag), max_read_len, C.byref(p))
    if rc != 0:
        raise BucketError(rc, L.ema_bucket_last_error().decode())
    return _take(L, p)


def parse_bucket(text: bytes, bc_len: int = 16, is_haplotag: bool = False, max_read_len: int = 255) -> Bucket:
    L = _lib()
    p = C.POINTER(_Bucket)()
    rc = L.ema_bucket_parse(text, len(text), bc_len, int(is_haplotag), max_read_len, C.byref(p))
    if rc != 0:
        raise BucketError(rc, L.ema_bucket_last_error().decode())
    return _take(L, p)


def encode_barcode(bc: bytes, is_haplotag: bool = False) -> int:
    L = _lib()
    v = C.c_uint64()
    rc = L.ema_barcode_encode(bc, len(bc), int(is_haplotag), C.byref(v))
    if rc != 0:
        raise BucketError(rc, "bad barcode")
    return v.value


def decode_barcode(bc: int, bc_len: int = 16, is_haplotag: bool = False) -> bytes:
    L = _lib()
    buf = C.create_string_buffer(40)
    L.ema_barcode_decode(bc, bc_len, int(is_haplotag), buf)
    return buf.raw[:12 if is_haplotag else bc_len]
