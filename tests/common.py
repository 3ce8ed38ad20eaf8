"""Shared helpers for the test-suite: small synthetic references built once per session."""
from __future__ import annotations

import os
import tempfile

import numpy as np

from ema_amd import synth, build_index

_CACHE = {}


def small_ref(kind="two_contigs"):
    """Builds (once) a small synthetic reference + index; returns (prefix, contigs)."""
    if kind in _CACHE:
        return _CACHE[kind]
    d = tempfile.mkdtemp(prefix="ema_ref_")
    if kind == "two_contigs":
        ctg = synth.make_genome([200000, 100000], seed=1)
    elif kind == "repeats":      # repeat-rich: exercises max_occ, chain filter, rescue
        ctg = synth.make_genome([600000, 300000, 50000], seed=7, short_rep=0.2, long_rep=0.1, segdup=0.05)
    elif kind == "ngaps":
        ctg = synth.make_genome([150000, 80000], seed=3, n_gaps=20)
    elif kind == "mid":          # a few Mbp for GPU throughput smoke tests
        ctg = synth.make_genome([3000000, 1000000], seed=11)
    else:
        raise KeyError(kind)
    prefix = os.path.join(d, "ref.fa")
    synth.write_fasta(prefix, ctg)
    build_index(prefix)
    _CACHE[kind] = (prefix, ctg)
    return _CACHE[kind]
