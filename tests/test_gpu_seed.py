"""GPU parity of the seeding kernels (SMEM / seed-interval collection) against the CPU oracle, through the C ABI:
K1 (one read per lane: the bulk) and K1w (one wavefront per read: the long reads of the full-capacity tier).  The debug
entry point runs on the full-capacity tier, whose seeding kernel the tuning knob full_seed_lane selects."""
import numpy as np
import pytest

import oracle_lib as O
from common import same_intervals, small_ref
from ema_amd import synth
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _check(kind, n_pairs, seed, kernel, tuning, pairs=None, min_seed_len=None, **kw):
    tuning(full_seed_lane="1" if kernel == "lane" else "0")
    prefix, ctg = small_ref(kind)
    if pairs is None:
        pairs = synth.make_pairs(ctg, n_pairs, seed=seed, **kw)
    eo = None
    if min_seed_len is not None:
        from ema_amd.engine import default_opts
        eo = default_opts()
        eo.min_seed_len = min_seed_len
    eng = Engine(prefix, opts=eo)
    table = eng.index_info()["kmer_k"] > 0 and kernel == "lane"      # K1 with the k-mer interval table: k' is not produced (0)
    eng.stage(pairs.bases, pairs.off)
    intv, n_intv = eng.debug_seeds()
    idx, opt = O.Index(prefix), O.default_opt()
    if min_seed_len is not None:
        opt.min_seed_len = min_seed_len
    bad = 0
    for r in range(2 * pairs.n):
        ref = O.collect_intv(idx, opt, pairs.read(r))
        got = [(int(v[3]) >> 32, int(v[3]) & 0xffffffff, int(v[0]), int(v[1]), int(v[2])) for v in intv[r, :n_intv[r]]]
        bad += not same_intervals(got, [tuple(int(t) for t in d) for d in ref], idx, table)
    eng.close()
    assert bad == 0, f"{bad} of {2 * pairs.n} reads have different seed intervals"


KERNELS = pytest.mark.parametrize("kernel", ["lane", "wave"])


@KERNELS
def test_seed_parity_clean(kernel, tuning):
    _check("two_contigs", 600, 21, kernel, tuning)


@KERNELS
def test_seed_parity_with_n_bases(kernel, tuning):
    _check("two_contigs", 300, 22, kernel, tuning, n_rate=0.01)


@KERNELS
def test_seed_parity_repeats(kernel, tuning):
    _check("repeats", 600, 23, kernel, tuning)


@KERNELS
def test_seed_parity_250bp(kernel, tuning):
    # config 5 of BASELINE.json (2x250 bp): beyond the reference's MAX_READ_LEN (include/align.h:61), supported here
    _check("repeats", 200, 24, kernel, tuning, len1=250, len2=250)


@pytest.mark.parametrize("k", [0, 5, 11, 12])
def test_seed_parity_kmer_table_depths(k, tuning):
    """K1 with the k-mer interval table at several depths (0: none, the build that produces k' too; 11, 12: levels in the packed
    part of the table although the test genome would get 8 by itself) and without parking rounds left out."""
    tuning(kmer_k=str(k))
    _check("repeats", 500, 25, "lane", tuning, sub_rate=0.02, n_rate=0.003)


@pytest.mark.parametrize("tail", ["1", "0"])
@pytest.mark.parametrize("k", ["", "5", "11"])
@pytest.mark.parametrize("kind", ["two_contigs", "ngaps"])
def test_seed_parity_text_tails_at_their_edges(kind, k, tail, tuning):
    """K1's text tails (a single-occurrence match followed along the 2-bit text instead of through rank queries) and pass 3's
    jump, where they must stop exactly as the rank queries do: across the strand junction, at the text's end, at an ambiguous
    base, at the read's end, beyond one load of text, in every word phase (tests/common.py, text_edge_pairs) -- with the tails
    on and off (tuning knob seed_tail) and at several table depths."""
    from common import text_edge_pairs
    tuning(seed_tail=tail)
    if k:
        tuning(kmer_k=k)
    _prefix, ctg = small_ref(kind)
    _check(kind, 0, 0, "lane", tuning, pairs=text_edge_pairs(ctg))


@pytest.mark.parametrize("tail", ["1", "0"])
def test_seed_parity_tails_on_and_off(tail, tuning):
    tuning(seed_tail=tail)
    _check("repeats", 600, 27, "lane", tuning, sub_rate=0.01, n_rate=0.002)
    _check("two_contigs", 300, 28, "lane", tuning, len1=250, len2=250, sub_rate=0.002)


@pytest.mark.parametrize("wtest", ["1", "0", "no-anchors"])
@pytest.mark.parametrize("k", ["", "5", "11"])
def test_seed_parity_window_test_of_pass_2(k, wtest, tuning):
    """Pass 2's window test (k_seed.hip: a re-seeding search is skipped when no min_seed_len-base window over its position can be
    frequent enough) on and off (seed_wtest), at several table depths: repeat-rich reads (searches that DO report), clean ones,
    ambiguous bases, 250 bp; with and without the anchors that build on it (a pass-1 SMEM found on the text and reported by position) --
    the intervals equal the oracle's either way."""
    tuning(seed_wtest="0" if wtest == "0" else "1")
    tuning(seed_anchor="0" if wtest == "no-anchors" else "1")      # (anchors: single-occurrence matches reported by position)
    if k:
        tuning(kmer_k=k)
    _check("repeats", 500, 31, "lane", tuning, sub_rate=0.01, n_rate=0.004)
    _check("two_contigs", 300, 32, "lane", tuning, len1=250, len2=250, sub_rate=0.003)
    _check("ngaps", 300, 33, "lane", tuning, n_rate=0.01)


@pytest.mark.parametrize("msl", [12, 25, 31])
def test_seed_parity_other_minimum_seed_lengths(msl, tuning):
    """bwa's -k: the window test, the anchors and pass 3's jump all take their lengths from min_seed_len (the windows are that
    long; the jump is min(kmer_k, min_seed_len) bases; re-seeding starts at 1.5 x) -- below, above and far above the default 19."""
    _check("repeats", 400, 34, "lane", tuning, min_seed_len=msl, sub_rate=0.01, n_rate=0.002)
    _check("two_contigs", 200, 35, "lane", tuning, min_seed_len=msl)


def test_pipeline_with_and_without_the_table(tuning):
    import test_gpu_pipeline as TP
    for k in ("0", "11"):
        tuning(kmer_k=k)
        TP._check("repeats", 600, 26, sub_rate=0.02)


@pytest.mark.parametrize("onepass", ["1", "0"])
@pytest.mark.parametrize("park,rounds", [(40, 4), (60, 8)])
def test_seed_parity_one_control_pass_per_tick_and_parking(onepass, park, rounds, tuning):
    """one control pass per tick (seed_flags bit 2, the default) leaves machines without a request on their lane, and a
    parked wave carries them with has_req = 0.  Both settings of the knob, with parking thresholds high enough (a wave parks as soon
    as it is down to `park` machines, up to `rounds` launches per series) that request-less machines ARE parked and resumed: repeat-rich
    reads (long chains of ticks), ambiguous bases (passes that end between two states), 250 bp."""
    tuning(seed_onepass=onepass, seed_park=park, seed_rounds=rounds)
    _check("repeats", 700, 36, "lane", tuning, sub_rate=0.01, n_rate=0.004)
    _check("two_contigs", 300, 37, "lane", tuning, len1=250, len2=250, n_rate=0.01)


@pytest.mark.parametrize("k,split", [("", "1"), ("0", "1"), ("5", "1"), ("11", "1"), ("", "0"), ("0", "0")])      # (the old form: with and without the table)
def test_seed_parity_pass_3_in_its_own_kernel_and_inside_k1(k, split, tuning):
    """Pass 3 (bwt_seed_strategy1, the LAST-like seeds) runs as a kernel of its own behind K1 (k_seed_p3.hip, the default) or inside
    K1's machine (tuning knob seed_split3=0, round 4's form): the same interval sets either way, with and without the k-mer table
    (the jump over a seed's first bases, the reverse-complement entry at the table's last level), ambiguous bases (starts moved
    behind them), repeat-rich reads (seeds that reach max_mem_intv late), 250 bp, other minimum seed lengths."""
    tuning(seed_split3=split)
    if k:
        tuning(kmer_k=k)
    _check("repeats", 500, 41, "lane", tuning, sub_rate=0.01, n_rate=0.004)
    _check("ngaps", 300, 42, "lane", tuning, n_rate=0.02)
    _check("two_contigs", 200, 43, "lane", tuning, len1=250, len2=250, sub_rate=0.003)
    _check("repeats", 300, 44, "lane", tuning, min_seed_len=12, sub_rate=0.01)
    _check("repeats", 300, 45, "lane", tuning, min_seed_len=25, sub_rate=0.01)


@pytest.mark.parametrize("split", ["1", "0"])
def test_lean_budget_runs_on_across_the_passes(split, tuning):
    """The lean tier's extend budget counts all three passes: with pass 3 in its own kernel the count K1 reached travels with the read
    (DevOpts::seed_ext), so a read over the budget in pass 3 is given up there and redone by the full-capacity tier -- the whole
    path against the oracle with a budget that about half the reads exceed, some of them only in pass 3."""
    import test_gpu_pipeline as TP
    from ema_amd.engine import default_opts
    tuning(seed_split3=split)
    o = default_opts()
    o.lean_seed_extends = 300
    TP._check("repeats", 500, 46, eopts=o, sub_rate=0.01)
    o2 = default_opts()
    o2.lean_seed_extends = 120
    TP._check("two_contigs", 300, 47, eopts=o2)
