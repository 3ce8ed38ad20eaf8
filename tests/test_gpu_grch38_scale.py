"""Parity at the BENCHMARK's own scale inside `pytest -m gpu` (BASELINE configs[1]): the 3.1 Gbp synthetic
reference bench.py measures on -- 20 contigs, 6.2 G suffix-array rows of 8 bytes, three rank superblocks, the k-mer table at depth
14, the 2-bit text of K1's tails at 1.55 GB: the branches only a human-size index takes, in the PRODUCT build -- and 112,000 read
pairs through the whole path (K1-K4, both capacity tiers) against the oracle, every read's candidate list (regions, positions, NM,
CIGARs) by digest (oracle/pair.c, orc_digest_pairs, on every CPU the box grants):

  * 80,000 pairs of the 10x mix the benchmark times (0.5 % substitutions, 0.05 % indels, 1 % chimeric);
  * 32,000 pairs of a rescue-heavy mix (6 % substitutions, ten times the indels, 5 % chimeric: mates that do not seed and are found
    by mem_matesw, long extensions, gapped final alignments).

Every pair the full-capacity tier redid is in the comparison (the digest covers the whole batch), and the test insists that some
were.  The reference and its index are built as bench.py builds them (native genome generator, suffix array sorted on the GPU) in
bench.py's own work directory, so a bench run on the same box reuses them."""
import os
import tempfile
import types

import numpy as np
import pytest

import oracle_lib as O
from ema_amd import synth
from ema_amd.engine import Engine, default_opts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_grch38_scale_reference_112k_pairs_against_the_oracle():
    import bench
    args = types.SimpleNamespace(genome_mbp=3100.0)
    workdir = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
    os.makedirs(workdir, exist_ok=True)
    bench.build_reference(args, workdir)
    lens, _name = bench.genome_spec(args)
    prefix = os.path.join(workdir, "ref.fa")
    flat = np.load(os.path.join(workdir, "genome.npy"), mmap_mode="r")
    ctg, at = [], 0
    for n in lens:
        ctg.append(np.asarray(flat[at:at + n])); at += n
    assert sum(lens) > 3_000_000_000
    mixes = [synth.make_pairs(ctg, 80000, seed=171), synth.make_pairs(ctg, 32000, seed=172, sub_rate=0.06, indel_rate=0.005, chimeric=0.05)]
    o = default_opts()
    o.batch_pairs = 81920
    eng = Engine(prefix, opts=o)
    info = eng.index_info()
    assert info["sa_width"] == 8 and info["n_super"] >= 2 and info["kmer_k"] == 14      # the human-size branches, not a small index
    idx, opt = O.Index(prefix), O.default_opt()
    n_threads = len(os.sched_getaffinity(0))
    n_redone = 0
    try:
        for pairs in mixes:
            batch = eng.align_pairs(pairs.bases, pairs.off)
            assert batch.status.max() == 0
            n_redone += int(batch.n_redone)
            got = O.cand_digest(batch.cand, batch.cigar, batch.cand_off)
            want, _ = O.digest_pairs(idx, opt, pairs.bases, pairs.off, n_threads)
            bad = np.nonzero(got != want)[0]
            assert len(bad) == 0, f"{len(bad)} of {2 * pairs.n} reads differ from the oracle, first: read {int(bad[0])}"
            assert int(batch.cand_off[-1]) > pairs.n      # (the mixes do align)
    finally:
        eng.close()
    assert n_redone > 100, "the full-capacity tier took no part in the comparison"
