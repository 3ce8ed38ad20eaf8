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
    names = None
    if kind == "two_contigs":
        ctg = synth.make_genome([200000, 100000], seed=1)
    elif kind == "repeats":      # repeat-rich: exercises max_occ, chain filter, rescue
        ctg = synth.make_genome([600000, 300000, 50000], seed=7, short_rep=0.2, long_rep=0.1, segdup=0.05)
    elif kind == "ngaps":
        ctg = synth.make_genome([150000, 80000], seed=3, n_gaps=20)
    elif kind == "tiny_repeats":  # 120 Kbp: 240 K BWT symbols = four 2^16-symbol rank superblocks in the ss16 test build
        ctg = synth.make_genome([80000, 40000], seed=17, short_rep=0.2, long_rep=0.1, segdup=0.05)
    elif kind == "exact_dups":    # exact copies of 3 kb segments: far apart (candidates of equal likelihood in different clouds: XA
        ctg = synth.make_genome([300000, 120000], seed=19, short_rep=0.0, long_rep=0.0, segdup=0.0)      # entries) and 15 kb apart
        rng = np.random.default_rng(19)      # (two candidates of one read inside one cloud: bad clouds)
        g = ctg[0]
        for k in range(12):
            src = 5000 + 22000 * k
            dst = src + (15000 if k % 3 == 0 else 140000) % (len(g) - 10000)
            dst = dst % (len(g) - 4000)
            g[dst:dst + 3000] = g[src:src + 3000]
    elif kind == "repeat_family":   # 640 diverged copies (2.5 %) of a 500 bp element, either strand, 800 bp apart: reads from a copy have
        ctg = synth.make_genome([530000], seed=23)      # dozens to hundreds of seed occurrences and chains (K2b's medium layout, and
        rng = np.random.default_rng(23)                 # beyond EMA_MED_CHAINS = 256 chains its move to the slab)
        g = ctg[0]
        elem = rng.integers(0, 4, 500).astype(np.uint8)
        for k in range(640):
            e = elem.copy()
            hit = rng.random(500) < 0.025
            e[hit] = (e[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
            if k & 1:
                e = (3 - e)[::-1]
            at = 3000 + 800 * k
            g[at:at + 500] = e
    elif kind == "tiny_family":   # 118 Kbp (four 2^16-symbol superblocks in the ss16 build) holding 440 diverged copies of a 200 bp element:
        ctg = synth.make_genome([118000], seed=31, short_rep=0.0, long_rep=0.0, segdup=0.0)      # reads from a copy have several hundred chains
        rng = np.random.default_rng(31)
        g = ctg[0]
        elem = rng.integers(0, 4, 200).astype(np.uint8)
        for k in range(440):
            e = elem.copy()
            hit = rng.random(200) < 0.015
            e[hit] = (e[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
            if k & 1:
                e = (3 - e)[::-1]
            at = 2000 + 260 * k
            g[at:at + 200] = e
    elif kind == "with_alt":      # two primary contigs + two ALT contigs (diverged copies of primary segments), named in <prefix>.alt
        ctg = synth.make_genome([200000, 100000], seed=29)
        rng = np.random.default_rng(29)
        for src, lo, n, div in ((0, 40000, 24000, 0.012), (1, 30000, 12000, 0.03)):
            a = ctg[src][lo:lo + n].copy()
            hit = rng.random(n) < div
            a[hit] = (a[hit] + rng.integers(1, 4, int(hit.sum()))) % 4
            ctg.append(a)
        names = ["chr1", "chr2", "chr1_alt1", "chr2_alt1"]
    elif kind == "mid":          # a few Mbp for GPU throughput smoke tests
        ctg = synth.make_genome([3000000, 1000000], seed=11)
    else:
        raise KeyError(kind)
    prefix = os.path.join(d, "ref.fa")
    synth.write_fasta(prefix, ctg, names=names)
    build_index(prefix)
    if kind == "with_alt":      # bwa.kit's .alt is SAM-formatted: a header, then one line per ALT contig starting with its name
        with open(prefix + ".alt", "w") as f:
            f.write("@SQ\tSN:chr1\tLN:200000\nchr1_alt1\t0\tchr1\t40001\t60\t24000M\t*\t0\t0\t*\t*\nchr2_alt1\t0\tchr2\t30001\t60\t12000M\t*\t0\t0\t*\t*\n")
    _CACHE[kind] = (prefix, ctg)
    return _CACHE[kind]


def golden_workload():
    """tests/golden/oracle_regression.json (see make_vectors.py there): the committed workload with its index rebuilt in a
    temp dir.  Returns (prefix, Pairs, expected intervals per read, expected candidates per pair and mate)."""
    import json
    if "golden" not in _CACHE:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_regression.json")
        with open(path) as f:
            doc = json.load(f)
        code = {"A": 0, "C": 1, "G": 2, "T": 3}
        ctg = [np.array([code[c] for c in s], dtype=np.uint8) for s in doc["contigs"]]
        d = tempfile.mkdtemp(prefix="ema_golden_")
        prefix = os.path.join(d, "g.fa")
        synth.write_fasta(prefix, ctg)
        build_index(prefix)
        reads = [r.encode() for r in doc["reads"]]
        off = np.zeros(len(reads) + 1, np.uint32)
        off[1:] = np.cumsum([len(r) for r in reads])
        pairs = synth.Pairs(np.frombuffer(b"".join(reads), dtype=np.uint8), off)
        _CACHE["golden"] = (prefix, pairs, doc["intervals"], doc["candidates"])
    return _CACHE["golden"]


def text_edge_pairs(ctg, seed=41):
    """Reads whose long single-occurrence matches run into the places where K1's text tails (k_seed.hip) must stop as the rank
    queries do: across the junction of the forward strand and its reverse complement, into the end of the text, into an
    ambiguous base, past 224 bases (more than one load of text), at the read's end; and starting in every word phase.
    Returns synth.Pairs (an even number of reads of assorted lengths, paired up arbitrarily)."""
    rng = np.random.default_rng(seed)
    g = np.concatenate(ctg)
    g = np.where(g > 3, 0, g).astype(np.uint8)      # as the index stores ambiguous reference bases (any base will do here)
    text = np.concatenate([g, (3 - g)[::-1]])
    n_text = len(text)
    reads = []

    def take(at, n):
        return text[at:at + n].copy()

    for d in range(0, 40):      # across the strand junction, every phase
        reads.append(take(len(g) - 100 - d, 180))
    for d in range(0, 40):      # up to the text's end, then bases that cannot match
        r = take(n_text - 90 - d, 90 + d)
        reads.append(np.concatenate([r, rng.integers(0, 4, 40, dtype=np.uint8)]))
    for d in range(0, 34):      # exact reads of every length class around one and two loads of text
        reads.append(take(5000 + 37 * d, 200 + d))
        reads.append(take(9000 + 41 * d, 255 - d))
    for d in range(0, 34):      # an ambiguous base right behind / inside a long match
        r = take(20000 + 53 * d, 150)
        r[60 + d] = 4
        reads.append(r)
        r = take(30000 + 59 * d, 150)
        r[149 - (d % 8)] = 4
        reads.append(r)
    for d in range(0, 34):      # one mismatch at every word phase
        r = take(40000 + 61 * d, 150)
        r[70 + d] = (r[70 + d] + 1) & 3
        reads.append(r)
    if len(reads) & 1:
        reads.append(take(50000, 100))
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    off = np.zeros(len(reads) + 1, np.uint32)
    off[1:] = np.cumsum([len(r) for r in reads])
    return synth.Pairs(lut[np.concatenate(reads)], off)


BYPOS = 1 << 63


def same_intervals(got, exp, idx, table_mode):
    """K1's intervals of one read against the oracle's: lists of (start, end, k, k', size).  In table mode K1 does not produce k'
    (0), and it may hand a single-occurrence interval over BY POSITION (k' = EMA_INTV_BYPOS, k = the occurrence's place in the
    text: what bwt_sa() returns for the oracle's row)."""
    if len(got) != len(exp):
        return False
    for g, e in zip(got, exp):
        if g[3] == BYPOS:
            if (g[0], g[1], g[4]) != (e[0], e[1], e[4]) or g[4] != 1 or g[2] != idx.sa(e[2]):
                return False
        elif (g[0], g[1], g[2], g[4]) != (e[0], e[1], e[2], e[4]) or g[3] != (0 if table_mode else e[3]):
            return False
    return True
