"""Deterministic synthetic inputs for the `ema align` hot path.

Neither a genome nor reads exist on the build or GPU boxes (no network), so the
benchmark and the tests generate both:

* `make_genome`  -- contigs of i.i.d. bases (41 % GC) with injected repeat
  families so that the engine's `max_occ` paths (reference src/align.c:185) are
  exercised: a short interspersed family, a long one and segmental duplications.
* `make_pairs`   -- linked-read style FR read pairs drawn from barcoded
  molecules, with substitutions, indels and a small chimeric fraction, in the
  layout of EMA's "special FASTQ" bucket lines
  (`BARCODE NAME READ1 QUAL1 READ2 QUAL2`, reference src/align.c:759-806).
"""
from __future__ import annotations

import numpy as np

GENOME_SEED = 0x454D41
READS_SEED = 0x31305831
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0, 4], dtype=np.uint8)


def _mutate(rng, seq, div):
    """Substitutions at rate `div` (indels in repeats are not needed to exercise seeding)."""
    out = seq.copy()
    hit = rng.random(len(seq)) < div
    n = int(hit.sum())
    if n:
        out[hit] = (out[hit] + rng.integers(1, 4, n, dtype=np.uint8)) & 3
    return out


def make_genome(contig_lens, seed=GENOME_SEED, short_rep=0.10, long_rep=0.05, segdup=0.01, n_gaps=0):
    """Returns a list of uint8 arrays (values 0..3, 4 = N), one per contig."""
    rng = np.random.default_rng(seed)
    total = int(sum(contig_lens))
    # in pieces: choice() returns 8-byte integers (25 GB at once for a human-size genome); the stream of draws is the same
    g = np.empty(total, dtype=np.uint8)
    for at in range(0, total, 1 << 26):
        n = min(1 << 26, total - at)
        g[at:at + n] = rng.choice(4, size=n, p=[0.295, 0.205, 0.205, 0.295])
    fam_short = rng.integers(0, 4, 300, dtype=np.uint8)
    fam_long = rng.integers(0, 4, 6000, dtype=np.uint8)

    def inject(consensus, frac, dlo, dhi):
        n_copies = int(total * frac / len(consensus))
        for _ in range(n_copies):
            L = len(consensus)
            if L > 1000:  # long family members are usually truncated
                L = int(rng.integers(500, len(consensus) + 1))
            s0 = int(rng.integers(0, len(consensus) - L + 1))
            copy = _mutate(rng, consensus[s0:s0 + L], rng.uniform(dlo, dhi))
            if rng.random() < 0.5:
                copy = (3 - copy)[::-1]
            p = int(rng.integers(0, total - L))
            g[p:p + L] = copy

    if total > 2 * 6000:
        inject(fam_short, short_rep, 0.10, 0.15)
        inject(fam_long, long_rep, 0.05, 0.20)
        n_sd = max(1, int(total * segdup / 30000)) if segdup > 0 else 0
        for _ in range(n_sd):
            L = int(rng.integers(min(10000, total // 8), min(100000, total // 4) + 1))
            src = int(rng.integers(0, total - L))
            dst = int(rng.integers(0, total - L))
            g[dst:dst + L] = _mutate(rng, g[src:src + L].copy(), rng.uniform(0.01, 0.02))
    for _ in range(n_gaps):
        L = int(rng.integers(10, 500))
        p = int(rng.integers(0, total - L))
        g[p:p + L] = 4
    out, off = [], 0
    for L in contig_lens:
        out.append(g[off:off + L])
        off += L
    return out


def make_genome_native(contig_lens, seed=GENOME_SEED, short_rep=0.10, long_rep=0.05, segdup=0.01):
    """The same model through libema_index.so's ema_synth_genome (csrc/synth_genome.cpp: its own generator, all host threads): a
    3.1 Gbp reference in a few seconds where make_genome takes 45.  bench.py's reference; the tests keep make_genome."""
    import ctypes as C
    import os
    L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libema_index.so"))
    L.ema_synth_genome.argtypes = [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_int]
    total = int(sum(contig_lens))
    g = np.empty(total, dtype=np.uint8)
    rc = L.ema_synth_genome(g.ctypes.data, total, seed, short_rep, long_rep, segdup, 0)
    if rc != 0:
        raise RuntimeError(f"ema_synth_genome failed ({rc})")
    out, off = [], 0
    for n in contig_lens:
        out.append(g[off:off + n])
        off += n
    return out


def write_fasta(path, contigs, names=None, width=60):
    names = names or [f"chr{i + 1}" for i in range(len(contigs))]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for name, c in zip(names, contigs):
            f.write(b">" + name.encode() + b"\n")
            txt = lut[c]
            n_full = len(txt) // width
            if n_full:
                body = np.empty((n_full, width + 1), dtype=np.uint8)
                body[:, :width] = txt[:n_full * width].reshape(n_full, width)
                body[:, width] = 10
                f.write(body.tobytes())
            if len(txt) % width:
                f.write(txt[n_full * width:].tobytes() + b"\n")
    return names


class Pairs:
    """A batch of read pairs: flat ASCII buffer + offsets (mate1 of pair i = read 2i, mate2 = read 2i+1)."""

    def __init__(self, bases, off, barcodes=None, truth=None):
        self.bases = bases            # uint8 ASCII
        self.off = off                # uint32, 2n+1
        self.barcodes = barcodes      # uint8 [n, 16] ASCII or None
        self.truth = truth            # dict of arrays or None

    @property
    def n(self):
        return (len(self.off) - 1) // 2

    def read(self, r):
        return self.bases[self.off[r]:self.off[r + 1]].tobytes()

    def take(self, ids):
        """The pairs with the given numbers, in that order, as a batch of their own."""
        ids = np.asarray(ids, dtype=np.int64)
        reads = np.stack([2 * ids, 2 * ids + 1], axis=1).reshape(-1)
        lo = self.off[reads].astype(np.int64)
        n = self.off[reads + 1].astype(np.int64) - lo
        off = np.zeros(len(reads) + 1, dtype=np.int64)
        off[1:] = np.cumsum(n)
        src = np.repeat(lo - off[:-1], n) + np.arange(int(off[-1]))
        return Pairs(np.asarray(self.bases)[src], off.astype(np.uint32), None if self.barcodes is None else self.barcodes[ids])

    def subset(self, lo, hi):
        o = self.off[2 * lo:2 * hi + 1]
        return Pairs(self.bases[o[0]:o[-1]].copy(), (o - o[0]).astype(np.uint32),
                     None if self.barcodes is None else self.barcodes[lo:hi])


def make_pairs(contigs, n_pairs, seed=READS_SEED, len1=127, len2=150, sub_rate=0.005, indel_rate=0.0005,
               chimeric=0.01, n_rate=0.0, pairs_per_barcode=200, flat=None):
    """Vectorised 10x-style simulator.  R1 is `len1` bases (150 - 16 barcode - 7 trim, reference
    cpp/correct.cc:550), R2 `len2`; FR orientation, outer insert ~ N(350, 60) clamped to [max(len)+20, 700]."""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.int64)
    g = flat if flat is not None else np.concatenate(contigs)      # flat: the contigs back to back (e.g. a memory map)
    offs = np.concatenate([[0], np.cumsum(lens)])
    L = max(len1, len2)
    # molecules: each barcode owns ~10 molecules of ~50 kb; reads are drawn from them
    n_bc = max(1, n_pairs // pairs_per_barcode)
    n_mol = n_bc * 10
    usable = lens > 2000
    p_ctg = (lens * usable) / (lens * usable).sum()
    mol_ctg = rng.choice(len(contigs), size=n_mol, p=p_ctg)
    mol_len = np.clip(rng.exponential(50000, n_mol), 10000, 200000).astype(np.int64)
    mol_len = np.minimum(mol_len, lens[mol_ctg] - 1)
    mol_start = (rng.random(n_mol) * (lens[mol_ctg] - mol_len)).astype(np.int64)
    pair_bc = np.sort(rng.integers(0, n_bc, n_pairs))
    pair_mol = pair_bc * 10 + rng.integers(0, 10, n_pairs)
    ins = np.clip(rng.normal(350, 60, n_pairs), L + 20, 700).astype(np.int64)
    ins = np.minimum(ins, mol_len[pair_mol] - 1)
    frag = mol_start[pair_mol] + (rng.random(n_pairs) * (mol_len[pair_mol] - ins)).astype(np.int64)
    frag_g = frag + offs[mol_ctg[pair_mol]]
    strand = rng.integers(0, 2, n_pairs)
    chim = rng.random(n_pairs) < chimeric

    def fetch(starts, length, rev):
        idx = starts[:, None] + np.arange(length)[None, :]
        m = g[idx]
        r = _COMP[m][:, ::-1]
        return np.where(rev[:, None].astype(bool), r, m)

    # forward-strand fragment: R1 = left end (+), R2 = revcomp of right end; flipped for strand 1
    left1 = fetch(frag_g, len1, np.zeros(n_pairs, dtype=np.int64))
    right2 = fetch(frag_g + ins - len2, len2, np.ones(n_pairs, dtype=np.int64))
    right1 = fetch(frag_g + ins - len1, len1, np.ones(n_pairs, dtype=np.int64))
    left2 = fetch(frag_g, len2, np.zeros(n_pairs, dtype=np.int64))
    r1 = np.where(strand[:, None] == 0, left1, right1)
    r2 = np.where(strand[:, None] == 0, right2, left2)
    if chim.any():
        nch = int(chim.sum())
        other = rng.integers(0, len(g) - len2 - 1, nch)
        r2[chim] = fetch(other, len2, rng.integers(0, 2, nch))

    def add_subs(m):
        hit = rng.random(m.shape) < sub_rate
        m = m.copy()
        sel = hit & (m < 4)
        m[sel] = (m[sel] + rng.integers(1, 4, int(sel.sum()), dtype=np.uint8)) & 3
        if n_rate > 0:
            m[rng.random(m.shape) < n_rate] = 4
        return m

    r1, r2 = add_subs(r1), add_subs(r2)

    def add_indels(m):
        # one indel event per affected read keeps the length fixed: delete k bases and pad from a
        # random tail, or insert k random bases and truncate
        n, Lr = m.shape
        ev = rng.random(n) < indel_rate * Lr
        for i in np.nonzero(ev)[0]:
            k = int(rng.geometric(0.5))
            p = int(rng.integers(10, Lr - 10 - k)) if Lr > 2 * (10 + k) + 1 else Lr // 2
            row = m[i]
            if rng.random() < 0.5:
                new = np.concatenate([row[:p], row[p + k:], rng.integers(0, 4, k, dtype=np.uint8)])
            else:
                new = np.concatenate([row[:p], rng.integers(0, 4, k, dtype=np.uint8), row[p:]])[:Lr]
            m[i] = new
        return m

    r1, r2 = add_indels(r1), add_indels(r2)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    inter = np.empty((n_pairs, len1 + len2), dtype=np.uint8)
    inter[:, :len1] = lut[r1]
    inter[:, len1:] = lut[r2]
    off = np.empty(2 * n_pairs + 1, dtype=np.uint32)
    base = np.arange(n_pairs, dtype=np.int64) * (len1 + len2)
    off[0:-1:2] = base
    off[1::2] = base + len1
    off[-1] = n_pairs * (len1 + len2)
    # barcodes: random 16-mers, never all-A (all-A encodes to the sentinel 0, reference src/align.c:1060)
    bcs = rng.integers(0, 4, (n_bc, 16), dtype=np.uint8)
    bcs[(bcs == 0).all(axis=1), 0] = 1
    barcodes = _ACGT[bcs][pair_bc]
    truth = {"contig": mol_ctg[pair_mol], "frag": frag, "ins": ins, "strand": strand, "chimeric": chim}
    return Pairs(inter.reshape(-1), off, barcodes, truth)


def write_special_fastq(path, pairs: Pairs, qual="F"):
    """EMA bucket file: `BC NAME R1 Q1 R2 Q2` per line (reference src/align.c:759-806, cpp/correct.cc:497-612)."""
    with open(path, "wb") as f:
        for i in range(pairs.n):
            r1, r2 = pairs.read(2 * i), pairs.read(2 * i + 1)
            f.write(b" ".join([pairs.barcodes[i].tobytes(), b"@s%d" % i, r1, qual.encode() * len(r1),
                               r2, qual.encode() * len(r2)]) + b"\n")


def write_special_fastq_fixed(path, pairs: Pairs, qual=b"F"):
    """The same file for a batch whose mates all have one length each (what make_pairs produces), written as ONE array:
    identifiers are zero-padded to a fixed width, so every line has the same length."""
    n = pairs.n
    l1 = int(pairs.off[1] - pairs.off[0]); l2 = int(pairs.off[2] - pairs.off[1])
    reads = np.asarray(pairs.bases).reshape(n, l1 + l2)
    w = max(1, len(str(n - 1)))
    ids = np.char.zfill(np.arange(n).astype(str), w).astype("S").view(np.uint8).reshape(n, w)
    L = 16 + 1 + 2 + w + 1 + l1 + 1 + l1 + 1 + l2 + 1 + l2 + 1
    out = np.empty((n, L), dtype=np.uint8)
    c = 0
    out[:, c:c + 16] = pairs.barcodes; c += 16
    out[:, c] = 32; out[:, c + 1] = ord("@"); out[:, c + 2] = ord("s"); c += 3
    out[:, c:c + w] = ids; c += w
    out[:, c] = 32; c += 1
    out[:, c:c + l1] = reads[:, :l1]; c += l1
    out[:, c] = 32; c += 1
    out[:, c:c + l1] = qual[0]; c += l1
    out[:, c] = 32; c += 1
    out[:, c:c + l2] = reads[:, l1:]; c += l2
    out[:, c] = 32; c += 1
    out[:, c:c + l2] = qual[0]; c += l2
    out[:, c] = 10
    out.tofile(path)


def bench_batch(job):
    """Worker of bench.py's read simulation: (genome.npy, contig lengths, pairs, seed, len1, len2, out.npz) -> one batch on disk.
    The genome is memory-mapped, so a pool of these shares one copy in the page cache."""
    gpath, lens, n_pairs, seed, len1, len2, out = job
    flat = np.load(gpath, mmap_mode="r")
    ctg, at = [], 0
    for n in lens:
        ctg.append(flat[at:at + n]); at += n
    p = make_pairs(ctg, n_pairs, seed=seed, len1=len1, len2=len2, flat=flat)
    tmp = out + ".tmp.npz"
    np.savez(tmp, bases=p.bases, off=p.off, barcodes=p.barcodes)
    import os
    os.replace(tmp, out)
    return out
