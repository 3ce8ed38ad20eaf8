"""ctypes mirror of include/ema_count.h: `ema count` (reference cpp/count.cc:38-182) -- barcode counts of an interleaved FASTQ
stream into <prefix>.ema-fcnt / <prefix>.ema-ncnt, byte-identical to the reference's files."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class CountStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("total_reads", "nice_reads", "ignored_reads", "bytes", "whitelist", "nice_barcodes", "full_blocks")]


def _L():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, os.environ.get("EMA_ENGINE_LIB", "libema_engine.so"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make`")
        L = C.CDLL(path)
        L.ema_count_fastq.restype = C.c_int
        L.ema_count_fastq.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t, C.c_int, C.POINTER(CountStats)]
        L.ema_count_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def count_fastq(whitelist: str | None, fastq, prefix: str, max_map_size: int = 1 << 30, is_haplotag: bool = False) -> dict:
    """fastq: a path or an open file descriptor (the reference reads stdin).  Returns the statistics; raises on an error code."""
    fd = os.open(fastq, os.O_RDONLY) if isinstance(fastq, str) else int(fastq)
    try:
        st = CountStats()
        rc = _L().ema_count_fastq(whitelist.encode() if whitelist else None, fd, prefix.encode(), max_map_size, int(is_haplotag), C.byref(st))
    finally:
        if isinstance(fastq, str):
            os.close(fd)
    if rc != 0:
        raise RuntimeError(f"ema_count_fastq failed ({rc}): {_L().ema_count_last_error().decode()}")
    return {n: int(getattr(st, n)) for n, _ in CountStats._fields_}
