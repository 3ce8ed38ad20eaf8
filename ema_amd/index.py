"""ctypes binding of the FM-index builder (`ema_amd/csrc/index_build.cpp`)."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libema_index.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        _lib = ctypes.CDLL(path)
        _lib.ema_index_build.argtypes = [ctypes.c_char_p, ctypes.c_int]
        _lib.ema_index_build.restype = ctypes.c_int
    return _lib


def build_index(fasta_path: str, n_threads: int = 0) -> None:
    """Writes <fasta>.{bwt,sa,fsa,pac,ann,amb,fai} next to the FASTA."""
    rc = _load().ema_index_build(fasta_path.encode(), n_threads)
    if rc != 0:
        raise RuntimeError(f"ema_index_build({fasta_path}) failed with code {rc}")
