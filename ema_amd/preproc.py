"""ctypes mirror of include/ema_preproc.h: `ema preproc` (reference cpp/correct.cc:271-633) -- barcode correction and bucketing of an
interleaved FASTQ stream into <dir>/ema-bin-NNN and <dir>/ema-nobc, byte-identical to the reference's files."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


class PreprocStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("no_change", "no_barcode", "h1_corrected", "h2_corrected", "corrected_strings", "pairs_written",
                                        "pairs_nobc", "pairs_skipped", "whitelist")]


def _L():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, os.environ.get("EMA_ENGINE_LIB", "libema_engine.so"))
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make`")
        L = C.CDLL(path)
        L.ema_preproc_fastq.restype = C.c_int
        L.ema_preproc_fastq.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.POINTER(PreprocStats)]
        L.ema_preproc_last_error.restype = C.c_char_p
        _lib = L
    return _lib


def preproc_fastq(whitelist: str | None, ncnt_paths, out_dir: str, fastq, do_h2: bool = False, buffer_size: int = 10 << 20,
                  do_bx_format: bool = False, n_threads: int = 1, n_buckets: int = 500, is_haplotag: bool = False) -> dict:
    """fastq: a path or an open file descriptor (the reference reads stdin).  Returns the statistics; raises on an error code."""
    fd = os.open(fastq, os.O_RDONLY) if isinstance(fastq, str) else int(fastq)
    try:
        arr = (C.c_char_p * len(ncnt_paths))(*[p.encode() for p in ncnt_paths])
        st = PreprocStats()
        rc = _L().ema_preproc_fastq(whitelist.encode() if whitelist else None, arr, len(ncnt_paths), out_dir.encode(), int(do_h2), buffer_size,
                                    int(do_bx_format), n_threads, n_buckets, int(is_haplotag), fd, C.byref(st))
    finally:
        if isinstance(fastq, str):
            os.close(fd)
    if rc != 0:
        raise RuntimeError(f"ema_preproc_fastq failed ({rc}): {_L().ema_preproc_last_error().decode()}")
    return {n: int(getattr(st, n)) for n, _ in PreprocStats._fields_}
