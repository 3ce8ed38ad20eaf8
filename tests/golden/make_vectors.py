"""Writes tests/golden/oracle_regression.json: a small fixed workload (reference contigs, read pairs) and what the
oracle (oracle/) returns for it -- seed intervals per read and the final candidate lists per mate.

These are REGRESSION vectors, not reference-pinned golden vectors: the reference (arshajii/ema) ships no tests or
fixtures for this path and its engine (the lh3/bwa submodule) is absent from the tree, so there is nothing upstream to
pin against (parity unpinned, see oracle/oracle.h and DESIGN.md).  They freeze the oracle's behaviour at the point where
the GPU path was verified bit-exact against it, so that a later change to either side shows up as a diff here.

Usage (from the repository root, after `make`):  python tests/golden/make_vectors.py
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from ema_amd import synth, build_index  # noqa: E402


def workload():
    ctg = synth.make_genome([60000, 25000], seed=21, short_rep=0.2, long_rep=0.1, segdup=0.05)
    pairs = synth.make_pairs(ctg, 40, seed=22, sub_rate=0.02, indel_rate=0.003, chimeric=0.1, n_rate=0.002)
    return ctg, pairs


def run(ctg, pairs):
    d = tempfile.mkdtemp(prefix="ema_golden_")
    prefix = os.path.join(d, "g.fa")
    synth.write_fasta(prefix, ctg)
    build_index(prefix)
    idx, opt = O.Index(prefix), O.default_opt()
    out = {"intervals": [], "candidates": []}
    for r in range(2 * pairs.n):
        out["intervals"].append([[int(v) for v in t] for t in O.collect_intv(idx, opt, pairs.read(r))])
    for p in range(pairs.n):
        res = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        out["candidates"].append([[{k: (float(np.float32(v)) if k == "frac_rep" else v) for k, v in c.items()} for c in mate] for mate in res])
    return out


if __name__ == "__main__":
    ctg, pairs = workload()
    doc = {"contigs": ["".join("ACGT"[b] for b in c) for c in ctg],
           "reads": [pairs.read(r).decode() for r in range(2 * pairs.n)]}
    doc.update(run(ctg, pairs))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes")
