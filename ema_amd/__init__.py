"""ema_amd -- MI355X-native engine for the `ema align` seed-and-extend hot path.

The package holds only what that path needs: the HIP kernels and C-ABI library
(`csrc/`, built in-tree as `libema_engine.so`), the host-side FM-index builder
(`libema_index.so`), a thin ctypes binding used by the tests and `bench.py`, and
the synthetic-input generators.  There is no CPU fallback: every entry point
fails loudly if the HIP library is missing or no GPU is present.
"""
from . import synth  # noqa: F401
from .index import build_index  # noqa: F401
