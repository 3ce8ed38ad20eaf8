"""CPU tests of the oracle and the index builder.  The reference ships no tests or golden vectors for this path
 and the engine's arithmetic lives in the absent lh3/bwa submodule, so the oracle is checked for
internal consistency against brute-force models: suffix-array search, exhaustive SMEM enumeration, scalar DP
re-scoring.  PARITY UNPINNED against real bwa."""
import ctypes as C

import numpy as np
import pytest

import dp_cases as D
import oracle_lib as O
from common import small_ref
from ema_amd import synth


@pytest.fixture(scope="module")
def tiny():
    import os, tempfile
    from ema_amd import build_index
    d = tempfile.mkdtemp(prefix="ema_tiny_")
    rng = np.random.default_rng(3)
    # low-complexity contigs so that repeats, ties and multi-occurrence intervals are common
    c1 = rng.integers(0, 4, 3000).astype(np.uint8)
    c1[1000:1400] = c1[200:600]
    c1[2000:2300] = (3 - c1[300:600])[::-1]
    c2 = np.tile(rng.integers(0, 4, 37).astype(np.uint8), 20)
    prefix = os.path.join(d, "t.fa")
    synth.write_fasta(prefix, [c1, c2])
    build_index(prefix)
    text = np.concatenate([c1, c2])
    both = np.concatenate([text, (3 - text)[::-1]])
    return prefix, [c1, c2], both


def test_index_files_against_naive_suffix_array(tiny):
    prefix, ctg, T = tiny
    n = len(T)
    s = bytes(T.tolist())
    order = sorted(range(n + 1), key=lambda i: s[i:])
    idx = O.Index(prefix)
    # every sampled and LF-walked SA value agrees with the naive suffix array
    for row in range(0, n + 1, 7):
        if row == 0:
            continue
        assert idx.sa(row) == order[row]
    # flat SA file
    raw = np.fromfile(prefix + ".fsa", dtype=np.uint8)
    assert bytes(raw[:8]) == b"EMAFSA01"
    sa = np.frombuffer(raw[24:].tobytes(), dtype=np.uint32)
    assert sa.tolist() == order
    # occ: cumulative counts of the BWT
    bwt = [T[i - 1] for i in order if i != 0]
    L = O.lib()
    primary = order.index(0)
    for k in list(range(0, n, 53)) + [n - 1]:
        kk = k - (1 if k >= primary else 0)
        for c in range(4):
            assert L.orc_occ(idx.h, k, c) == sum(1 for b in bwt[:kk + 1] if b == c)


def brute_smems(T, q, min_len=19):
    """All supermaximal exact matches of q (ACGT-only stretches) against text T, with occurrence counts."""
    s = bytes(T.tolist())
    n = len(q)

    def occ(a, b):
        pat = bytes(q[a:b].tolist())
        cnt, start = 0, 0
        while True:
            i = s.find(pat, start)
            if i < 0:
                return cnt
            cnt += 1
            start = i + 1

    mems = []
    for a in range(n):
        if q[a] > 3:
            continue
        b = a
        while b < n and q[b] < 4 and occ(a, b + 1) > 0:
            b += 1
        if b > a:
            mems.append((a, b))
    smems = [m for m in mems if not any(o != m and o[0] <= m[0] and m[1] <= o[1] for o in mems)]
    return sorted((a, b, occ(a, b)) for a, b in set(smems) if b - a >= min_len)


def test_pass1_smems_against_brute_force(tiny):
    prefix, ctg, T = tiny
    idx, opt = O.Index(prefix), O.default_opt()
    opt.max_mem_intv = 0           # pass 3 off
    opt.split_factor = 1e6         # pass 2 off (no SMEM is long enough to be re-seeded)
    rng = np.random.default_rng(9)
    text = np.concatenate(ctg)
    for _ in range(25):
        p = int(rng.integers(0, len(text) - 120))
        q = D.mutate(rng, text[p:p + 110], sub=0.03, indel=0.005, n_rate=0.01)
        if rng.random() < 0.5:
            q = (3 - np.minimum(q, 3))[::-1].astype(np.uint8)
        got = O.collect_intv(idx, opt, bytes(b"ACGTN"[x] for x in q))
        got = sorted((a, b, x2) for a, b, x0, x1, x2 in got)
        assert got == brute_smems(T, q)


def test_intervals_point_at_real_occurrences(tiny):
    prefix, ctg, T = tiny
    idx, opt = O.Index(prefix), O.default_opt()
    rng = np.random.default_rng(10)
    text = np.concatenate(ctg)
    s = bytes(T.tolist())
    for _ in range(20):
        p = int(rng.integers(0, len(text) - 160))
        q = D.mutate(rng, text[p:p + 150], sub=0.02, indel=0.0)
        for a, b, x0, x1, x2 in O.collect_intv(idx, opt, bytes(b"ACGTN"[x] for x in q)):
            pat = bytes(q[a:b].tolist())
            rc = bytes((3 - q[a:b])[::-1].tolist())
            for k in range(int(x2)):
                pos = idx.sa(x0 + k)
                assert s[pos:pos + b - a] == pat
                pos = idx.sa(x1 + k)
                assert s[pos:pos + b - a] == rc


def test_introsort_restatement_sorts_and_is_deterministic():
    rng = np.random.default_rng(11)
    L = O.lib()
    for n in [0, 1, 2, 3, 16, 17, 18, 100, 1000, 5000]:
        a = rng.integers(0, 50, n).astype(np.uint64)     # many ties
        b = a.copy()
        L.orc_introsort_u64(n, a.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert (np.diff(a.astype(np.int64)) >= 0).all() and sorted(b.tolist()) == a.tolist()
    # adversarial input for the depth limit (comb-sort fallback): organ-pipe and sawtooth patterns
    for n in [4097, 10000]:
        a = np.concatenate([np.arange(n // 2), np.arange(n - n // 2)[::-1]]).astype(np.uint64)
        L.orc_introsort_u64(n, a.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert (np.diff(a.astype(np.int64)) >= 0).all()


def affine_score(ops, q, t, opt):
    """Re-scores a CIGAR with the affine model; returns None if it does not consume both sequences."""
    x = y = sc = 0
    for op in ops:
        o, l = op & 0xf, op >> 4
        if o == 0:
            for i in range(l):
                sc += opt.mat[int(t[y + i]) * 5 + int(q[x + i])]
            x += l; y += l
        elif o == 1:
            sc -= opt.o_ins + opt.e_ins * l; x += l
        elif o == 2:
            sc -= opt.o_del + opt.e_del * l; y += l
    return sc if (x, y) == (len(q), len(t)) else None


def test_global_alignment_traceback_rescoring():
    rng = np.random.default_rng(12)
    opt = O.default_opt()
    qs, ts, ws = D.global_cases(rng, 300, max_q=120)
    for q, t, w in zip(qs, ts, ws):
        sc, ops = D.oracle_global(q, t, w)
        assert affine_score(ops, q, t, opt) == sc


def naive_local(q, t, opt):
    H = np.zeros((len(t) + 1, len(q) + 1), dtype=np.int64)
    E = np.zeros_like(H); F = np.zeros_like(H)
    best = 0
    for i in range(1, len(t) + 1):
        for j in range(1, len(q) + 1):
            E[i, j] = max(E[i - 1, j] - opt.e_del, H[i - 1, j] - opt.o_del - opt.e_del, 0)
            F[i, j] = max(F[i, j - 1] - opt.e_ins, H[i, j - 1] - opt.o_ins - opt.e_ins, 0)
            H[i, j] = max(0, H[i - 1, j - 1] + opt.mat[int(t[i - 1]) * 5 + int(q[j - 1])], E[i, j], F[i, j])
            best = max(best, H[i, j])
    return int(best)


def test_local_score_against_textbook_gotoh():
    rng = np.random.default_rng(13)
    opt = O.default_opt()
    qs, ts, prm = D.local_cases(rng, 25, max_q=60)
    for q, t, p in zip(qs, ts, prm):
        t = t[:150]
        assert D.oracle_local_pass(q, t, 16, 19, 0x10000)[0] == naive_local(q, t, opt)


def test_extension_of_identical_sequences():
    opt = O.default_opt()
    rng = np.random.default_rng(14)
    for n in [1, 5, 50, 200]:
        q = rng.integers(0, 4, n).astype(np.uint8)
        r = D.oracle_extend(q, np.concatenate([q, rng.integers(0, 4, 20).astype(np.uint8)]), [100, 5, 100, 30])
        assert r[0] == 30 + n and r[1] == n and r[2] == n and r[4] == 30 + n   # score, qle, tle, gscore


def test_pairs_map_back_to_their_origin():
    """Property at the bridge level: clean FR pairs come back at the simulated coordinates, both mates, proper strands."""
    prefix, ctg = small_ref("two_contigs")
    idx, opt = O.Index(prefix), O.default_opt()
    pairs = synth.make_pairs(ctg, 60, seed=15, sub_rate=0.002, indel_rate=0.0, chimeric=0.0)
    ok = 0
    for p in range(pairs.n):
        m1, m2 = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        if not m1 or not m2:
            continue
        tr = pairs.truth
        frag, ins, strand, contig = int(tr["frag"][p]), int(tr["ins"][p]), int(tr["strand"][p]), int(tr["contig"][p])
        a, b = m1[0], m2[0]
        left = a if strand == 0 else b
        right = b if strand == 0 else a
        if left["rid"] == contig and left["pos"] == frag and left["is_rev"] == 0 and right["is_rev"] == 1 and \
                right["pos"] + (right["re"] - right["rb"]) == frag + ins:
            ok += 1
    assert ok >= 55


def test_oracle_reproduces_the_committed_regression_vectors():
    """tests/golden/oracle_regression.json freezes the oracle at the point where the GPU path was verified against it
    (regression vectors -- the reference has none to pin against)."""
    from common import golden_workload
    prefix, pairs, intervals, candidates = golden_workload()
    idx, opt = O.Index(prefix), O.default_opt()
    for r in range(2 * pairs.n):
        assert [[int(v) for v in t] for t in O.collect_intv(idx, opt, pairs.read(r))] == intervals[r]
    for p in range(pairs.n):
        res = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        got = [[{k: (float(np.float32(v)) if k == "frac_rep" else v) for k, v in c.items()} for c in mate] for mate in res]
        assert got == candidates[p]


def test_candidate_digest_of_arrays_equals_the_oracles():
    """oracle_lib.cand_digest (numpy, from candidate arrays: what bench.py's large spot check applies to the engine's output)
    against orc_digest_pairs (C, from the oracle's own candidates)."""
    import numpy as np
    from common import small_ref
    from ema_amd import synth
    from ema_amd.engine import CAND_DTYPE
    prefix, ctg = small_ref("repeats")
    pairs = synth.make_pairs(ctg, 120, seed=77, sub_rate=0.02)
    idx, opt = O.Index(prefix), O.default_opt()
    want, _ = O.digest_pairs(idx, opt, pairs.bases, pairs.off, 4)
    rows, pool, read_off = [], [], [0]
    for p in range(pairs.n):
        ref = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            for d in ref[m]:
                r = np.zeros(1, dtype=CAND_DTYPE)
                for f in ("rb", "re", "qb", "qe", "score", "pos", "NM", "is_rev", "csub", "seedcov"):
                    r[f] = d[f]
                r["n_cigar"] = len(d["cigar"]); r["cigar_off"] = len(pool)
                pool.extend(d["cigar"])
                rows.append(r)
            read_off.append(len(rows))
    cand = np.concatenate(rows)
    got = O.cand_digest(cand, np.array(pool, dtype=np.uint32), np.array(read_off))
    assert got.tolist() == want.tolist()
    cand["NM"][len(cand) // 2] += 1      # any field of any candidate moves its read's digest
    assert (O.cand_digest(cand, np.array(pool, dtype=np.uint32), np.array(read_off)) != want).sum() == 1
