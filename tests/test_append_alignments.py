"""The host stage right behind the engine (reference src/align.c:846-911, 959-1061: append_alignments' filters, the
approximate mapq, the alignment likelihoods, the `unique` flag) -- the C ABI's ema_batch_append_alignments against the
oracle's restatement.  Host arithmetic only: runs without a GPU on a batch assembled from the oracle's candidates; the
GPU suite (test_gpu_pipeline.py) runs it on the engine's own batches."""
import numpy as np

import oracle_lib as O
from common import small_ref
from ema_amd import synth
from ema_amd import engine as E


def batch_from_oracle(prefix, pairs):
    idx, opt = O.Index(prefix), O.default_opt()
    cand_off, cands, cigar = [0], [], []
    for p in range(pairs.n):
        res = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            for d in res[m]:
                c = np.zeros((), dtype=E.CAND_DTYPE)
                for f in O.REG_FIELDS:
                    c[f] = d[f]
                c["pos"], c["is_rev"], c["NM"], c["n_cigar"], c["cigar_off"] = d["pos"], d["is_rev"], d["NM"], len(d["cigar"]), len(cigar)
                c["aln_score"], c["aln_sub"] = d["score"], max(d["sub"], d["csub"])
                cigar.extend(d["cigar"])
                cands.append(c)
            cand_off.append(len(cands))
    cand = np.array(cands, dtype=E.CAND_DTYPE) if cands else np.zeros(0, E.CAND_DTYPE)
    return E.Batch(np.array(cand_off, np.uint64), cand, np.array(cigar, np.uint32), np.zeros(2 * pairs.n, np.int32))


def check(prefix, pairs, batch):
    idx, opt = O.Index(prefix), O.default_opt()
    rec, pair_off = E.append_alignments(batch, pairs.off)
    assert len(pair_off) == pairs.n + 1 and pair_off[-1] == len(rec)
    n_unique = 0
    for p in range(pairs.n):
        exp = O.append_alignments(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        got = rec[pair_off[p]:pair_off[p + 1]]
        assert len(got) == len(exp), p
        for g, e in zip(got, exp):
            assert int(g["pair"]) == p and int(g["mate"]) == e["mate"]
            assert int(g["cand"]) - int(batch.cand_off[2 * p + e["mate"]]) == e["cand"]
            for f in ("clip", "clip_edit_dist", "mapq", "score_mapq", "unique"):
                assert int(g[f]) == e[f], (p, f)
            assert float(g["score"]) == e["score"], p      # same operations in the same order: bit-identical doubles
            n_unique += e["unique"]
    return len(rec), n_unique


def test_append_alignments_matches_the_oracle():
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 120, seed=61, sub_rate=0.03, indel_rate=0.004, chimeric=0.15, n_rate=0.002)
    n, n_unique = check(prefix, pairs, batch_from_oracle(prefix, pairs))
    assert n > pairs.n and 0 < n_unique < n      # multi-candidate reads, filtered candidates and unique reads all occur


def test_append_alignments_threaded_equals_serial(monkeypatch):
    """Above 4096 pairs the stage runs on several host threads (chunks of pairs laid end to end): same records."""
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 100, seed=63, sub_rate=0.03, chimeric=0.1)
    one = batch_from_oracle(prefix, pairs)
    reps = 50
    n_c, n_g = len(one.cand), len(one.cigar)
    cand = np.tile(one.cand, reps)
    cand["cigar_off"] = (cand["cigar_off"].astype(np.int64) + np.repeat(np.arange(reps) * n_g, n_c)).astype(np.uint32)
    cand_off = np.concatenate([one.cand_off[:-1].astype(np.int64) + k * n_c for k in range(reps)] + [np.array([reps * n_c])]).astype(np.uint64)
    big = E.Batch(cand_off, cand, np.tile(one.cigar, reps), np.zeros(2 * pairs.n * reps, np.int32))
    lens = np.diff(pairs.off.astype(np.int64))
    off = np.concatenate([[0], np.cumsum(np.tile(lens, reps))]).astype(np.uint32)
    monkeypatch.setenv("EMA_HOST_THREADS", "1")
    rec1, po1 = E.append_alignments(big, off)
    monkeypatch.setenv("EMA_HOST_THREADS", "7")
    rec7, po7 = E.append_alignments(big, off)
    assert (po1 == po7).all() and rec1.tobytes() == rec7.tobytes()
    base, _ = E.append_alignments(one, pairs.off)
    assert len(rec1) == reps * len(base) and (rec1["mapq"][:len(base)] == base["mapq"]).all()
    assert (rec1["pair"][len(base):2 * len(base)] == base["pair"] + pairs.n).all()
