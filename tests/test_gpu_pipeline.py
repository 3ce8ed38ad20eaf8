"""End-to-end GPU parity of the hot path through the C ABI: for every pair, the candidate lists of both mates
(regions after mate rescue, in the reference's order) with position, strand, NM and CIGAR must be identical to
the oracle's restatement of bwa_mem_mate_sw + bwa_smith_waterman (reference src/bwabridge.c:204-311)."""
import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine, default_opts

pytestmark = pytest.mark.gpu
FIELDS = [f for f in O.REG_FIELDS]


def compare(prefix, pairs, batch, opt=None):
    idx = O.Index(prefix)
    opt = opt if opt is not None else O.default_opt()
    bad = []
    for p in range(pairs.n):
        ref = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            got = []
            for c in batch.mate(p, m):
                d = {f: (float(c[f]) if f == "frac_rep" else int(c[f])) for f in FIELDS}
                d.update(pos=int(c["pos"]), is_rev=int(c["is_rev"]), NM=int(c["NM"]), cigar=batch.cigar_of(c).tolist())
                got.append(d)
            rf = []
            for d in ref[m]:
                d = dict(d)
                d["frac_rep"] = float(np.float32(d["frac_rep"]))
                rf.append(d)
            if rf != got:
                bad.append((p, m))
    return bad


def _check(kind, n_pairs, seed, eopts=None, oopt=None, **kw):
    prefix, ctg = small_ref(kind)
    pairs = synth.make_pairs(ctg, n_pairs, seed=seed, **kw)
    eng = Engine(prefix, opts=eopts)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0
    bad = compare(prefix, pairs, batch, oopt)
    assert not bad, f"{len(bad)} of {2 * pairs.n} reads differ from the oracle, first {bad[:5]}"


def test_pipeline_clean():
    _check("two_contigs", 1500, 41)


def test_pipeline_noisy_reads_trigger_rescue():
    _check("two_contigs", 800, 42, sub_rate=0.09, indel_rate=0.003)


def test_pipeline_few_mismatches():
    """2 % substitutions, no indels: most reads have 1-4 mismatches, i.e. extensions and final alignments right at the
    limits of the known-outcome shortcuts (one mismatch on the diagonal; equal spans with <= 3 mismatches are all-M)."""
    _check("two_contigs", 800, 51, sub_rate=0.02, indel_rate=0.0)


def test_pipeline_other_scoring():
    """Non-default match/mismatch/gap scores (asymmetric gap costs): the known-outcome shortcuts carry conditions on the
    scoring parameters, and every DP takes them from the options."""
    eo = default_opts()
    oo = O.default_opt()
    for o in (eo, oo):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = 2, 3, 5, 2, 4, 2
    for i in range(5):
        for j in range(5):
            oo.mat[i * 5 + j] = -1 if i == 4 or j == 4 else (oo.a if i == j else -oo.b)
    _check("two_contigs", 500, 53, eopts=eo, oopt=oo, sub_rate=0.02, indel_rate=0.002)
    # mismatches cheaper than the shortcuts' conditions allow (a + b >= the cheapest gap): the DPs must run
    eo2, oo2 = default_opts(), O.default_opt()
    for o in (eo2, oo2):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = 1, 7, 2, 1, 2, 1
    for i in range(5):
        for j in range(5):
            oo2.mat[i * 5 + j] = -1 if i == 4 or j == 4 else (oo2.a if i == j else -oo2.b)
    _check("two_contigs", 300, 54, eopts=eo2, oopt=oo2, sub_rate=0.02, indel_rate=0.002)


def test_pipeline_250bp():
    """config 5 of BASELINE.json (2x250 bp): beyond the reference's MAX_READ_LEN (include/align.h:61), supported here up to
    255; exercises the widest column layouts of the DPs."""
    _check("repeats", 300, 52, len1=250, len2=250, sub_rate=0.01, indel_rate=0.002)


@pytest.mark.parametrize("attempts,regions", [(1, 1), (3, 2), (0, 0)])
def test_set_aside_routes_of_rescue_and_final_alignment(attempts, regions, tuning):
    """K3t / K3r (one rescue attempt per wavefront, replayed in order) and K4t / K4r (one region per wavefront, laid into the CIGAR pool
    in order) normally take only pairs with >= 8 candidate anchors / reads with >= 8 regions left; with the thresholds at 1 every
    rescued pair and every read with a gapped region goes that way, at 0 none does.  Noisy reads (rescues in both directions,
    several regions per read) and a repeat-rich reference."""
    tuning(heavy_attempts=attempts, heavy_regions=regions)
    _check("two_contigs", 700, 46, sub_rate=0.08, indel_rate=0.004)
    _check("repeats", 500, 47, sub_rate=0.03, indel_rate=0.003, chimeric=0.1)


def test_pipeline_chimeric_and_n():
    _check("two_contigs", 600, 43, chimeric=0.3, n_rate=0.004)


def test_pipeline_repeats():
    _check("repeats", 600, 44)


def test_pipeline_repeat_family():
    """The 640-copy diverged repeat end to end (lean tier, full-capacity tier for the reads that outgrow it, rescue, CIGARs)."""
    _check("repeat_family", 200, 45)


def test_pipeline_empty_and_ragged_batches():
    prefix, ctg = small_ref("two_contigs")
    eng = Engine(prefix)
    # empty batch
    b = eng.align_pairs(np.zeros(0, np.uint8), np.zeros(1, np.uint32))
    assert len(b.cand) == 0 and len(b.cand_off) == 1
    # ragged: reads shorter than the seed length, 1-base reads, all-N reads next to normal ones
    pairs = synth.make_pairs(ctg, 8, seed=45)
    reads = [pairs.read(i) for i in range(16)]
    reads[1] = reads[1][:10]
    reads[2] = b"A"
    reads[5] = b"N" * 60
    reads[6] = reads[6][:19]
    reads[9] = b""
    off = np.zeros(17, np.uint32)
    off[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    batch = eng.align_pairs(bases, off)
    eng.close()
    rag = synth.Pairs(bases, off)
    assert not compare(prefix, rag, batch)


@pytest.mark.parametrize("pipeline", ["1", "0"])
def test_more_pairs_than_the_batch_capacity(pipeline, tuning):
    """ema_engine_align_pairs takes a whole bucket: beyond the engine's batch capacity it works in pieces (alternating
    over two sets of batch buffers, or over one with the tuning knob align_pipeline=0) and lays the results end to end; the split
    form (stage) still refuses what does not fit."""
    tuning(align_pipeline=pipeline)
    prefix, ctg = small_ref("two_contigs")
    o = default_opts()
    o.batch_pairs = 96
    eng = Engine(prefix, opts=o)
    pairs = synth.make_pairs(ctg, 500, seed=46, sub_rate=0.02, indel_rate=0.002)
    with pytest.raises(RuntimeError):
        eng.stage(pairs.bases, pairs.off)
    batch = eng.align_pairs_any(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0 and len(batch.cand_off) == 2 * pairs.n + 1
    assert not compare(prefix, pairs, batch)


def test_two_capacity_tiers_give_the_same_candidates():
    """Lean capacities small enough that many pairs overflow them: those pairs are redone on the device by the
    full-capacity tier and spliced back; the batch must still equal the oracle read for read."""
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 700, seed=47, sub_rate=0.03)
    o = default_opts()
    o.lean_intervals, o.lean_regions, o.lean_cigar_ops = 9, 2, 6
    o.n_streams = 3
    eng = Engine(prefix, opts=o)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    assert batch.status.max() == 0
    assert 10 < batch.n_redone < pairs.n
    assert not compare(prefix, pairs, batch)
    # the same engine, the same batch, three passes queued back to back without waiting in between
    eng.stage(pairs.bases, pairs.off)
    for _ in range(3):
        eng.run()
    again = eng.fetch()
    eng.close()
    assert again.n_redone == batch.n_redone
    assert (again.cand_off == batch.cand_off).all() and (again.cand == batch.cand).all() and (again.cigar == batch.cigar).all()


def test_full_tier_overflow_fails_loudly():
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 300, seed=48, sub_rate=0.03)
    o = default_opts()
    o.lean_intervals, o.lean_regions, o.lean_cigar_ops = 9, 2, 6
    o.full_tier_pairs = 4
    eng = Engine(prefix, opts=o)
    eng.stage(pairs.bases, pairs.off)
    eng.run()
    with pytest.raises(RuntimeError, match="full-capacity tier"):
        eng.fetch()
    b = eng.fetch(allow_limit=True)       # the batch is still returned: flagged reads carry status bits and no candidates
    eng.close()
    assert b.n_redone == 4 and (b.status != 0).sum() >= 2
    flagged = np.nonzero(b.status)[0]
    assert all(b.cand_off[r + 1] == b.cand_off[r] for r in flagged)


def test_lean_seeding_budget_long_reads_by_wave_or_full_tier(tuning):
    """A read whose seeding needs more FM-index extends than K1's lean budget is given up there and seeded again in place by K1w
    (one wavefront per read, no budget) -- or, with the tuning knob seed_long_wave=0 and beyond the room of the list of such reads, redone by the
    full-capacity tier: same candidates as the oracle on every route."""
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 500, seed=49)
    redone = {}
    for route in ("1", "0"):
        tuning(seed_long_wave=route)
        o = default_opts()
        o.lean_seed_extends = 250          # about the median read (of K1's requests: tails, window tests and anchors count one each)
        eng = Engine(prefix, opts=o)
        batch = eng.align_pairs(pairs.bases, pairs.off)
        eng.close()
        assert batch.status.max() == 0
        assert not compare(prefix, pairs, batch)
        redone[route] = batch.n_redone
    assert 40 < redone["0"] < pairs.n          # every long read's pair went through the full tier
    assert redone["1"] < redone["0"] // 4      # seeded in place: what is left are the pairs over a lean capacity


def test_long_reads_beyond_the_list_go_to_the_full_tier():
    """The list of long reads holds an eighth of a slice's reads (at least 1024): with nearly every read over a tiny budget the rest
    keep their flag and take the full tier's route, up to its capacity."""
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 4000, seed=50)
    o = default_opts()
    o.lean_seed_extends = 60
    o.batch_pairs = 4000
    o.n_streams = 1
    o.full_tier_pairs = 4000
    eng = Engine(prefix, opts=o)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0
    assert batch.n_redone > 1000
    assert not compare(prefix, pairs, batch)


def test_committed_regression_vectors():
    """The workload and expected candidate lists of tests/golden/oracle_regression.json, through the C ABI."""
    from common import golden_workload
    prefix, pairs, _, candidates = golden_workload()
    eng = Engine(prefix)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0
    for p in range(pairs.n):
        for m in range(2):
            got = []
            for c in batch.mate(p, m):
                d = {f: (float(c[f]) if f == "frac_rep" else int(c[f])) for f in FIELDS}
                d.update(pos=int(c["pos"]), is_rev=int(c["is_rev"]), NM=int(c["NM"]), cigar=batch.cigar_of(c).tolist())
                got.append(d)
            assert got == candidates[p][m], (p, m)


def test_append_alignments_on_engine_batches():
    """ema_batch_append_alignments (the host stage behind the engine: reference src/align.c:986-1061) on what the engine
    returned, against the oracle's candidates put through the oracle's restatement of the same stage."""
    from test_append_alignments import check
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 300, seed=62, sub_rate=0.03, indel_rate=0.004, chimeric=0.15, n_rate=0.002)
    eng = Engine(prefix)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    n, n_unique = check(prefix, pairs, batch)
    assert n > pairs.n and n_unique > 0


def test_second_engine_sharing_the_index():
    """ema_engine_open_shared: own buffers and streams, the first engine's index; both give the oracle's candidates, also
    when driven from two host threads at once."""
    import threading
    prefix, ctg = small_ref("two_contigs")
    first = Engine(prefix)
    second = Engine(None, share=first)
    pa = synth.make_pairs(ctg, 400, seed=64)
    pb = synth.make_pairs(ctg, 400, seed=65, sub_rate=0.03)
    out = {}

    def work(name, eng, pairs):
        out[name] = eng.align_pairs(pairs.bases, pairs.off)

    th = [threading.Thread(target=work, args=("a", first, pa)), threading.Thread(target=work, args=("b", second, pb))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    second.close()
    first.close()
    assert not compare(prefix, pa, out["a"]) and not compare(prefix, pb, out["b"])


def test_large_batch_properties():
    """At a batch size the oracle cannot check pair by pair in a test: (1) a random sample of pairs equals the oracle;
    (2) the result of a pair does not depend on where in the batch it sits or how the batch is sliced -- the same pairs
    in reverse order, on an engine with a different slice count and a different lean tier, give the same candidates."""
    prefix, ctg = small_ref("mid")
    n = 120000
    pairs = synth.make_pairs(ctg, n, seed=66)
    eng = Engine(prefix)
    a = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert a.status.max() == 0
    rng = np.random.default_rng(7)
    pick = np.sort(rng.choice(n, 600, replace=False))      # (at this size the kernels claim their work items four at a time)
    sub_reads = [pairs.read(2 * int(p) + m) for p in pick for m in range(2)]
    so = np.zeros(len(sub_reads) + 1, np.uint32)
    so[1:] = np.cumsum([len(r) for r in sub_reads])
    sample = synth.Pairs(np.frombuffer(b"".join(sub_reads), dtype=np.uint8), so)

    class View:      # the sampled pairs of batch `a`, renumbered
        def mate(self, p, m):
            return a.mate(int(pick[p]), m)

        def cigar_of(self, c):
            return a.cigar_of(c)
    assert not compare(prefix, sample, View())
    # reversed order, 2 slices, small lean capacities
    order = np.arange(n)[::-1]
    lens = np.diff(pairs.off.astype(np.int64))
    rl = np.stack([lens[2 * order], lens[2 * order + 1]], axis=1).ravel()
    roff = np.zeros(2 * n + 1, np.uint32)
    roff[1:] = np.cumsum(rl)
    rbases = np.empty(int(roff[-1]), np.uint8)
    src0 = pairs.off[2 * order].astype(np.int64)
    plen = lens[2 * order] + lens[2 * order + 1]
    dst0 = roff[0:2 * n:2].astype(np.int64)
    for s0, d0, ln in zip(src0.tolist(), dst0.tolist(), plen.tolist()):
        rbases[d0:d0 + ln] = pairs.bases[s0:s0 + ln]
    o = default_opts()
    o.n_streams = 2
    o.lean_intervals, o.lean_regions, o.lean_cigar_ops = 12, 3, 9
    eng = Engine(prefix, opts=o)
    b = eng.align_pairs(rbases, roff)
    eng.close()
    assert b.status.max() == 0 and b.n_redone > 0
    na = np.diff(a.cand_off.astype(np.int64)).reshape(n, 2)
    nb = np.diff(b.cand_off.astype(np.int64)).reshape(n, 2)
    assert (na == nb[::-1]).all()
    drop = ["cigar_off"]
    fa = a.cand[[f for f in a.cand.dtype.names if f not in drop]]
    fb = b.cand[[f for f in b.cand.dtype.names if f not in drop]]
    for p in rng.choice(n, 4000, replace=False).tolist():
        q = n - 1 - p
        for m in range(2):
            ca, cb = a.mate(p, m), b.mate(q, m)
            assert (fa[int(a.cand_off[2 * p + m]):int(a.cand_off[2 * p + m + 1])] == fb[int(b.cand_off[2 * q + m]):int(b.cand_off[2 * q + m + 1])]).all()
            for x, y in zip(ca, cb):
                assert (a.cigar_of(x) == b.cigar_of(y)).all()


@pytest.mark.gpu
def test_bucket_file_to_candidates(tmp_path):
    """A bucket file through the reader and the engine (include/ema_ingest.h -> include/ema_engine.h), against the
    oracle's reader and the oracle's aligner on the same file: same pairs in the same order, same candidates."""
    import random
    from ema_amd import ingest
    prefix, ctg = small_ref("two_contigs")
    pairs = synth.make_pairs(ctg, 400, seed=52, sub_rate=0.01, indel_rate=0.001, pairs_per_barcode=7)
    path = str(tmp_path / "bucket.fq")
    synth.write_special_fastq(path, pairs)
    lines = open(path, "rb").read().splitlines(keepends=True)
    random.Random(3).shuffle(lines)      # preproc writes buckets unsorted; the reader orders them by barcode
    open(path, "wb").write(b"".join(lines))
    bucket = ingest.read_bucket(path)
    want, groups = O.read_special_fastq(path)
    assert bucket.n_pairs == pairs.n == len(want) and len(groups) == len(bucket.group_off) - 1 > 20
    for i, (bc, ident, r1, q1, r2, q2) in enumerate(want):
        assert int(bucket.bc[i]) == bc and bucket.ident(i) == ident and bucket.read(2 * i) == r1 and bucket.read(2 * i + 1) == r2
    eng = Engine(prefix)
    batch = eng.align_pairs_any(bucket.bases, bucket.off)
    eng.close()
    assert batch.status.max() == 0
    ordered = synth.Pairs(bucket.bases, bucket.off)
    assert not compare(prefix, ordered, batch)


def test_chr20_scale_reference_50k_pairs_against_the_oracle():
    """Parity at the scale of BASELINE configs[0]'s reference: a 64.4 Mbp synthetic chromosome (suffix array built
    on the GPU), 32,000 pairs of the 10x mix and 20,000 of a rescue-heavy mix (6 % substitutions, ten times the indels, 5 % chimeric:
    mates that do not seed and are found by mem_matesw, long extensions, gapped final alignments) -- every read's candidate list
    (regions, positions, NM, CIGARs) against the oracle's, digest against digest (oracle/pair.c, orc_digest_pairs, on every CPU the
    box grants)."""
    import os
    import numpy as np
    from common import _CACHE
    import tempfile
    from ema_amd import build_index
    if "chr20" not in _CACHE:
        ctg = synth.make_genome([64_444_167], seed=synth.GENOME_SEED)
        prefix = os.path.join(tempfile.mkdtemp(prefix="ema_chr20_"), "chr20.fa")
        synth.write_fasta(prefix, ctg, names=["chr20"])
        build_index(prefix)
        _CACHE["chr20"] = (prefix, ctg)
    prefix, ctg = _CACHE["chr20"]
    mixes = [synth.make_pairs(ctg, 32000, seed=71), synth.make_pairs(ctg, 20000, seed=72, sub_rate=0.06, indel_rate=0.005, chimeric=0.05)]
    o = default_opts()
    o.batch_pairs = 32768
    eng = Engine(prefix, opts=o)
    idx, opt = O.Index(prefix), O.default_opt()
    n_threads = len(os.sched_getaffinity(0))
    try:
        for pairs in mixes:
            batch = eng.align_pairs(pairs.bases, pairs.off)
            assert batch.status.max() == 0
            got = O.cand_digest(batch.cand, batch.cigar, batch.cand_off)
            want, _ = O.digest_pairs(idx, opt, pairs.bases, pairs.off, n_threads)
            bad = np.nonzero(got != want)[0]
            assert len(bad) == 0, f"{len(bad)} of {2 * pairs.n} reads differ from the oracle, first: read {int(bad[0])}"
            assert int(batch.cand_off[-1]) > pairs.n      # (the mixes do align)
    finally:
        eng.close()


def test_extension_under_other_scorings_and_a_narrow_band():
    """The extension DPs of K2 away from the defaults (the cases round 5 wrote for its lane-per-seed executor, K2x, which measured
    slower than the wave DP and is gone -- DESIGN 3; they hold the wave route to the same bar): clean and noisy reads (indels: gapped
    paths, z-drop, dead extensions in repeats), 250 bp reads, ambiguous bases, other scorings (asymmetric gaps; mismatches cheaper than a
    gap, so the known-outcome shortcuts do not apply), a narrow band (w = 4: max_off >= 3/4 w doubles the band), and tiny lean
    capacities (the full-capacity tier runs the same route).  Same candidates as the oracle."""
    _check("two_contigs", 900, 141, sub_rate=0.03, indel_rate=0.004)
    _check("repeats", 700, 142, sub_rate=0.02, indel_rate=0.002, chimeric=0.05)
    _check("repeats", 300, 143, len1=250, len2=250, sub_rate=0.015, indel_rate=0.003)
    _check("ngaps", 400, 144, n_rate=0.01, sub_rate=0.01)
    eo, oo = default_opts(), O.default_opt()
    for o in (eo, oo):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = 2, 3, 5, 2, 4, 2
    for i in range(5):
        for j in range(5):
            oo.mat[i * 5 + j] = -1 if i == 4 or j == 4 else (oo.a if i == j else -oo.b)
    _check("two_contigs", 400, 145, eopts=eo, oopt=oo, sub_rate=0.02, indel_rate=0.003)
    eo2, oo2 = default_opts(), O.default_opt()
    for o in (eo2, oo2):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = 1, 7, 2, 1, 2, 1
    for i in range(5):
        for j in range(5):
            oo2.mat[i * 5 + j] = -1 if i == 4 or j == 4 else (oo2.a if i == j else -oo2.b)
    _check("two_contigs", 300, 146, eopts=eo2, oopt=oo2, sub_rate=0.02, indel_rate=0.002)
    eo3, oo3 = default_opts(), O.default_opt()
    eo3.w = oo3.w = 4
    _check("two_contigs", 400, 147, eopts=eo3, oopt=oo3, sub_rate=0.02, indel_rate=0.01)
    eo4 = default_opts()
    eo4.lean_intervals, eo4.lean_regions, eo4.lean_cigar_ops, eo4.full_tier_pairs = 10, 2, 8, 1024
    _check("repeats", 600, 148, eopts=eo4, sub_rate=0.02, indel_rate=0.002)
