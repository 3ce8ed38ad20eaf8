"""BASELINE configs[2]'s shape on one GPU: many bucket files streamed through reader -> engine -> append stage by ONE C-ABI
call (include/ema_stream.h, ema_stream_buckets), two sets of batch buffers on one index taking alternate buckets, results
delivered in order -- every bucket compared with the oracle's reader, aligner and append stage (reference
src/main.c:396-406 -x loop, src/align.c:258,307-349,986-1061)."""
import random

import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import stream, synth
from ema_amd.engine import Engine, default_opts
from test_gpu_pipeline import compare

pytestmark = pytest.mark.gpu


def _write_buckets(tmp_path, ctg, sizes, seed0):
    paths = []
    for k, n in enumerate(sizes):
        path = str(tmp_path / f"ema-bin-{k:03d}")
        if n == 0:
            open(path, "wb").close()
        else:
            pairs = synth.make_pairs(ctg, n, seed=seed0 + k, sub_rate=0.01, indel_rate=0.001, pairs_per_barcode=9)
            synth.write_special_fastq(path, pairs)
            lines = open(path, "rb").read().splitlines(keepends=True)
            random.Random(k).shuffle(lines)      # preproc writes buckets unsorted
            open(path, "wb").write(b"".join(lines))
        paths.append(path)
    return paths


def _check_stream(tmp_path, kind, sizes, batch_pairs, n_engines, seed0):
    prefix, ctg = small_ref(kind)
    paths = _write_buckets(tmp_path, ctg, sizes, seed0)
    o = default_opts()
    o.batch_pairs = batch_pairs
    eng = Engine(prefix, opts=o)
    so = stream.default_opts()
    so.n_engines = n_engines
    seen = []
    idx, opt = O.Index(prefix), O.default_opt()

    def sink(k, bucket, batch, rec, pair_off):
        assert k == len(seen), "buckets must arrive in input order"
        bases, off, bc = bucket
        want, groups = O.read_special_fastq(paths[k]) if sizes[k] else ([], [])
        assert len(want) == sizes[k] == len(bc)
        for i, (wbc, _ident, r1, _q1, r2, _q2) in enumerate(want):
            assert int(bc[i]) == wbc and bases[off[2 * i]:off[2 * i + 1]].tobytes() == r1 and bases[off[2 * i + 1]:off[2 * i + 2]].tobytes() == r2
        ordered = synth.Pairs(bases, off)
        assert batch.status.max(initial=0) == 0
        bad = compare(prefix, ordered, batch)
        assert not bad, f"bucket {k}: {len(bad)} reads differ from the oracle, first {bad[:4]}"
        # append stage: the records of a sample of pairs against the oracle's append_alignments
        assert len(pair_off) == ordered.n + 1 and int(pair_off[-1]) == len(rec)
        for p in list(range(0, ordered.n, max(1, ordered.n // 60))):
            exp = O.append_alignments(idx, opt, ordered.read(2 * p), ordered.read(2 * p + 1))
            got = rec[int(pair_off[p]):int(pair_off[p + 1])]
            assert len(got) == len(exp)
            for g, e in zip(got, exp):
                assert int(g["mate"]) == e["mate"] and int(g["cand"]) - int(batch.cand_off[2 * p + e["mate"]]) == e["cand"]
                assert (int(g["clip"]), int(g["mapq"]), int(g["score_mapq"]), int(g["unique"])) == (e["clip"], e["mapq"], e["score_mapq"], e["unique"])
                assert float(g["score"]) == e["score"]
        seen.append((len(groups), int(batch.cand_off[-1]), len(rec)))

    stats = stream.stream_buckets(eng, paths, sink, so)
    eng.close()
    assert len(seen) == len(sizes)
    for k, st in enumerate(stats):
        assert st["pairs"] == sizes[k] and st["rc"] == 0 and st["capacity_flags"] == 0
        assert (st["barcode_groups"], st["candidates"], st["records"]) == seen[k]
        assert sum(st["mapq_hist"]) == st["records"]


def test_ten_buckets_streamed_two_buffer_sets(tmp_path):
    """10 bucket files, one of them empty, back to back on the engine and its peer (alternate buckets: the older schedule)."""
    _check_stream(tmp_path, "two_contigs", [300, 220, 0, 410, 150, 380, 90, 260, 330, 120], 512, 2, 700)


def test_ten_buckets_streamed_passes_queued_two_deep(tmp_path):
    """The default schedule: one set of batch buffers, ema_engine_run_async -- layout and packing on the device, the next pass
    queued before the previous one is fetched, staging on its own thread."""
    _check_stream(tmp_path, "repeats", [300, 220, 0, 410, 150, 380, 90, 260, 330, 120, 0, 500], 512, 1, 705)


def test_async_passes_with_tiny_lean_capacities(tmp_path):
    """Lean capacities so small that a third of the pairs go through the full-capacity tier in every pass queued two deep."""
    prefix, ctg = small_ref("repeats")
    paths = _write_buckets(tmp_path, ctg, [400, 350, 300, 450], 730)
    o = default_opts()
    o.batch_pairs = 512
    o.lean_intervals, o.lean_regions, o.lean_cigar_ops = 12, 3, 9
    eng = Engine(prefix, opts=o)
    redone = []

    def sink(k, bucket, batch, rec, pair_off):
        bases, off, bc = bucket
        assert batch.status.max(initial=0) == 0
        assert not compare(prefix, synth.Pairs(bases, off), batch)
        redone.append(batch.n_redone)
        assert sorted(set(batch.redone.tolist())) == sorted(batch.redone.tolist()) and len(batch.redone) == batch.n_redone
    stream.stream_buckets(eng, paths, sink)
    eng.close()
    assert len(redone) == 4 and min(redone) > 5


def test_buckets_beyond_the_batch_capacity_in_the_stream(tmp_path):
    """Capacity 256 pairs: some buckets fit a batch, some go through the piece pipeline (both buffer sets at once)."""
    _check_stream(tmp_path, "repeats", [200, 700, 130, 1000, 256, 257, 40, 600], 256, 2, 720)


def test_buckets_beyond_the_batch_capacity_in_the_async_stream(tmp_path):
    """The same with passes queued two deep: a big bucket drains the pipeline and goes through ema_engine_align_pairs."""
    _check_stream(tmp_path, "repeats", [200, 700, 130, 1000, 256, 257, 40, 600], 256, 1, 725)


def test_stream_with_empty_buckets_at_the_ends(tmp_path):
    _check_stream(tmp_path, "two_contigs", [0, 200, 0, 0, 310, 128, 64, 500, 77, 0], 512, 1, 740)


def test_stream_batches_from_memory_equals_align_pairs():
    """ema_stream_batches (what bench.py times at the boundary) on distinct in-memory batches = one call per batch."""
    prefix, ctg = small_ref("two_contigs")
    batches = [synth.make_pairs(ctg, n, seed=760 + i, sub_rate=0.01) for i, n in enumerate((400, 350, 512, 60, 300, 450))]
    o = default_opts()
    o.batch_pairs = 512
    eng = Engine(prefix, opts=o)
    got = {}
    stream.stream_batches(eng, [(p.bases, p.off) for p in batches], lambda k, _b, batch, rec, po: got.__setitem__(k, (batch, rec, po)))
    for k, p in enumerate(batches):
        one = eng.align_pairs(p.bases, p.off)
        b = got[k][0]
        assert (one.cand_off == b.cand_off).all() and len(one.cand) == len(b.cand)
        for f in one.cand.dtype.names:      # field by field: the records' padding bytes are not part of the result; cigar_off is an offset
            if f != "cigar_off":            # into the batch's own CIGAR array -- a bucket cut out of a shared pass is a view into the pass's
                assert (one.cand[f] == b.cand[f]).all(), (k, f)
        assert all(one.cigar_of(x).tolist() == b.cigar_of(y).tolist() for x, y in zip(one.cand, b.cand))
    eng.close()


def test_a_bad_bucket_stops_the_stream_with_its_name(tmp_path):
    prefix, ctg = small_ref("two_contigs")
    paths = _write_buckets(tmp_path, ctg, [100, 100, 100], 780)
    open(paths[1], "ab").write(b"ACGT only_two_fields\n")
    eng = Engine(prefix)
    seen = []
    with pytest.raises(RuntimeError) as ei:
        stream.stream_buckets(eng, paths, lambda k, *_: seen.append(k))
    eng.close()
    assert seen == [0] and "ema-bin-001" in str(ei.value)


@pytest.mark.parametrize("tiny_lean", [False, True])
def test_device_side_batch_layout_equals_the_host_assembly(tiny_lean, tuning):
    """ema_engine_fetch_ticket with the batch laid out on the device (k_pack.hip, ema_launch_merge: slices + full tier -> one set in
    read order, downloaded into a page-locked buffer that is the batch) against round 3's assembly on the host (tuning knob device_merge=0):
    the same arrays, entry for entry -- with the default lean capacities and with tiny ones, where a third of the pairs come from the
    full tier's set; batches of assorted sizes so that passes share and split slices."""
    prefix, ctg = small_ref("repeats")
    batches = [synth.make_pairs(ctg, n, seed=790 + i, sub_rate=0.01) for i, n in enumerate((700, 64, 1024, 333, 1, 900))]
    res = {}
    for mode in ("1", "0", "small"):
        # "small": a merged set too small for the batches (room for 4,096 candidates and operations in all) -- the fetch must fall
        # back to the host assembly from the slices' own sets, not refuse the batch (a repeat-heavy bucket aborted the stream)
        tuning(device_merge="0" if mode == "0" else "1", merged_cand=0 if mode == "small" else None, merged_cigar=0 if mode == "small" else None)
        o = default_opts()
        o.batch_pairs = 1024
        if tiny_lean:
            o.lean_intervals, o.lean_regions, o.lean_cigar_ops, o.full_tier_pairs = 10, 2, 8, 1024
        eng = Engine(prefix, opts=o)
        got = {}
        stream.stream_batches(eng, [(p.bases, p.off) for p in batches], lambda k, _b, batch, rec, po: got.__setitem__(k, (batch, rec, po)))
        eng.close()
        res[mode] = got
    for k, p, first in [(k, p, f) for k, p in enumerate(batches) for f in ("1", "small")]:
        a, b = res[first][k], res["0"][k]
        assert (a[0].cand_off == b[0].cand_off).all() and (a[0].status == b[0].status).all()
        assert sorted(a[0].redone.tolist()) == sorted(b[0].redone.tolist())      # (the list's order is the order of the collecting atomics)
        for f in a[0].cand.dtype.names:
            if f != "cigar_off":
                assert (a[0].cand[f] == b[0].cand[f]).all(), (k, f)
        assert all(a[0].cigar_of(x).tolist() == b[0].cigar_of(y).tolist() for x, y in zip(a[0].cand, b[0].cand))
        assert (a[1] == b[1]).all() and (a[2] == b[2]).all()      # the append stage's records
        if tiny_lean and p.n > 100:
            assert a[0].n_redone > p.n // 20
    assert not compare(prefix, batches[0], res["1"][0][0])
