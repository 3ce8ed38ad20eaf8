"""The suffix array built on the GPU (ema_amd/csrc/k_sa.hip: two-base chunks, stable radix sort on 32-base keys, tied rows refined
round by round) gives the SAME index files as the host builder, byte for byte: .bwt, .sa, .fsa (and everything else ema_index_build
writes).  References with long exact repeats (hundreds of refinement rounds over a few rows), repeat families (many rows tied for a
few rounds), ambiguous bases, several contigs, and 8-byte rows."""
import filecmp
import os
import shutil
import tempfile

import numpy as np
import pytest

from ema_amd import build_index, synth

pytestmark = pytest.mark.gpu
EXTS = (".bwt", ".sa", ".fsa", ".pac", ".ann", ".amb", ".fai")


def _both(ctg, monkeypatch, sa64=False):
    d = tempfile.mkdtemp(prefix="ema_gpusa_")
    out = {}
    for mode in ("0", "1"):
        sub = os.path.join(d, mode)
        os.mkdir(sub)
        fa = os.path.join(sub, "ref.fa")
        synth.write_fasta(fa, ctg)
        monkeypatch.setenv("EMA_INDEX_GPU", mode)
        monkeypatch.setenv("EMA_INDEX_PROF", "1")
        if sa64:
            monkeypatch.setenv("EMA_INDEX_SA64", "1")
        build_index(fa)
        out[mode] = fa
    for e in EXTS:
        assert filecmp.cmp(out["0"] + e, out["1"] + e, shallow=False), e
    shutil.rmtree(d)


def test_random_and_repeat_rich_references(monkeypatch, capfd):
    _both(synth.make_genome([200000, 100000], seed=1), monkeypatch)
    _both(synth.make_genome([600000, 300000, 50000], seed=7, short_rep=0.2, long_rep=0.1, segdup=0.05), monkeypatch)
    err = capfd.readouterr().err
    assert "sa (gpu)" in err and "sa (host)" in err      # both builders really ran


def test_long_exact_repeats_and_homopolymers(monkeypatch):
    ctg = synth.make_genome([300000, 120000], seed=19, short_rep=0.0, long_rep=0.0, segdup=0.0)
    g = ctg[0]
    for k in range(12):      # exact 3 kb copies: ~190 rounds over their rows (forward strand and reverse complement)
        src = 5000 + 22000 * k
        dst = (src + (15000 if k % 3 == 0 else 140000)) % (len(g) - 4000)
        g[dst:dst + 3000] = g[src:src + 3000]
    g[250000:252500] = 0                     # a run of 2,500 A (and of T on the other strand): every suffix inside ties with its neighbours
    ctg[1][-700:] = 0                        # the text's forward half ends in A's; its reverse-complement half ends wherever contig 1 starts
    ctg[0][:900] = 3                         # ... with T's complemented: the whole text ENDS in 900 A's (exhausted suffixes tied with longer ones)
    _both(ctg, monkeypatch)


def test_ambiguous_bases_and_eight_byte_rows(monkeypatch):
    _both(synth.make_genome([150000, 80000], seed=3, n_gaps=20), monkeypatch)
    _both(synth.make_genome([80000, 40000], seed=17, short_rep=0.2, long_rep=0.1, segdup=0.05), monkeypatch, sa64=True)
