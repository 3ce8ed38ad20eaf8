"""GPU parity of the three wave DPs (extension, global + traceback, local pass) against the oracle's
scalar restatements of ksw_extend2 / ksw_global2 / ksw_u8+ksw_i16, through the C ABI.  Bit-exact."""
import numpy as np
import pytest

import dp_cases as D
from common import small_ref
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    prefix, _ = small_ref("two_contigs")
    e = Engine(prefix)
    yield e
    e.close()


def test_extend_parity(eng):
    rng = np.random.default_rng(101)
    qs, ts, prm = D.extend_cases(rng, 3000)
    # edge cases: 1-base query/target, h0 = 1, N-only query, zdrop off, tiny band
    qs += [np.array([2], np.uint8), np.array([4] * 30, np.uint8), np.arange(255, dtype=np.uint8) & 3]
    ts += [np.array([2], np.uint8), np.array([1] * 40, np.uint8), np.arange(300, dtype=np.uint8) & 3]
    prm = np.concatenate([prm, np.array([[100, 5, 100, 1], [100, 5, 100, 30], [3, 5, 0, 19]], np.int32)])
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    out, _ = eng.debug_dp(0, qb, qo, tb, to, prm)
    bad = [i for i in range(len(qs)) if D.oracle_extend(qs[i], ts[i], prm[i]) != out[i].tolist()]
    assert not bad, f"{len(bad)} extension tasks differ, first {bad[:5]}"


def test_global_parity(eng):
    rng = np.random.default_rng(102)
    qs, ts, prm = D.global_cases(rng, 2000)
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    out, cig = eng.debug_dp(1, qb, qo, tb, to, prm)
    bad = []
    for i in range(len(qs)):
        sc, ops = D.oracle_global(qs[i], ts[i], prm[i])
        if sc != out[i, 0] or ops != cig[i, :out[i, 1]].tolist():
            bad.append(i)
    assert not bad, f"{len(bad)} global tasks differ, first {bad[:5]}"


def test_local_parity(eng):
    rng = np.random.default_rng(103)
    qs, ts, prm = D.local_cases(rng, 1500)
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    out, _ = eng.debug_dp(2, qb, qo, tb, to, prm)
    bad = [i for i in range(len(qs))
           if D.oracle_local_pass(qs[i], ts[i], int(prm[i, 0]), int(prm[i, 1]), int(prm[i, 2])) != out[i].tolist()]
    assert not bad, f"{len(bad)} local tasks differ, first {bad[:5]}"


def test_local_stop_pass_parity(eng):
    """The reverse pass of ksw_align2 runs with XSTOP|score: stops at the first row reaching it."""
    rng = np.random.default_rng(104)
    qs, ts, prm = D.local_cases(rng, 500)
    first, _ = eng.debug_dp(2, *D.flat(qs), *D.flat(ts), prm)
    prm2 = prm.copy()
    prm2[:, 1] = 0x10000
    prm2[:, 2] = np.maximum(first[:, 0], 1)
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    out, _ = eng.debug_dp(2, qb, qo, tb, to, prm2)
    bad = [i for i in range(len(qs))
           if D.oracle_local_pass(qs[i], ts[i], int(prm2[i, 0]), 0x10000, int(prm2[i, 2])) != out[i].tolist()]
    assert not bad, f"{len(bad)} local XSTOP tasks differ, first {bad[:5]}"


@pytest.mark.parametrize("sc", [(2, 3, 5, 2, 4, 2), (1, 7, 2, 1, 2, 1), (5, 4, 9, 3, 12, 2)])
def test_dp_parity_other_scorings(sc):
    """The three DPs under other -A -B -O -E: asymmetric gap costs; mismatches dearer than the known-outcome shortcuts' conditions
    allow (a + b >= the cheapest gap: the row loops must run); and a match score of 5, where a 250-base query reaches 1250 + h0 --
    far beyond a byte, the range bwa's own 8-bit kernel leaves to ksw_i16 (the local pass is asked for both widths).
    Queries up to 250 bases (the widest column layout)."""
    prefix, _ = small_ref("two_contigs")
    eo, oo = D.scoring(*sc)
    e = Engine(prefix, opts=eo)
    try:
        rng = np.random.default_rng(200 + sc[0] * 7 + sc[1])
        qs, ts, prm = D.extend_cases(rng, 1200)
        prm[:, 3] = rng.integers(1, 150 * sc[0], len(prm))      # h0 up to a whole read of matches
        out, _ = e.debug_dp(0, *D.flat(qs), *D.flat(ts), prm)
        bad = [i for i in range(len(qs)) if D.oracle_extend(qs[i], ts[i], prm[i], oo) != out[i].tolist()]
        assert not bad, f"{len(bad)} extension tasks differ, first {bad[:5]}"
        qs, ts, prm = D.global_cases(rng, 800)
        out, cig = e.debug_dp(1, *D.flat(qs), *D.flat(ts), prm)
        bad = []
        for i in range(len(qs)):
            s_, ops = D.oracle_global(qs[i], ts[i], prm[i], oo)
            if s_ != out[i, 0] or ops != cig[i, :out[i, 1]].tolist():
                bad.append(i)
        assert not bad, f"{len(bad)} global tasks differ, first {bad[:5]}"
        qs, ts, prm = D.local_cases(rng, 600)
        if sc[0] * 250 >= 250:      # ksw_align2 picks the 16-bit kernel when qlen * max score >= 250 (oracle/dp.c)
            prm[[len(q) * sc[0] >= 250 for q in qs], 0] = 8
        out, _ = e.debug_dp(2, *D.flat(qs), *D.flat(ts), prm)
        bad = [i for i in range(len(qs))
               if D.oracle_local_pass(qs[i], ts[i], int(prm[i, 0]), int(prm[i, 1]), int(prm[i, 2]), oo) != out[i].tolist()]
        assert not bad, f"{len(bad)} local tasks differ, first {bad[:5]}"
    finally:
        e.close()


def test_global_band_layout_and_traceback_runs(eng):
    """The global DP with its lanes across the band and the traceback that takes a run of matches as one step (dev_dp.hpp), at
    their edges: bands 0-31, queries of 1-255 bases, long clean diagonals, gaps, ambiguous bases -- score and CIGAR against
    ksw_global2's restatement."""
    rng = np.random.default_rng(105)
    qs, ts, prm = D.band_cases(rng, 3000)
    out, cig = eng.debug_dp(1, *D.flat(qs), *D.flat(ts), prm, cigar_cap=640)
    bad = []
    for i in range(len(qs)):
        sc, ops = D.oracle_global(qs[i], ts[i], prm[i])
        if sc != out[i, 0] or ops != cig[i, :max(0, out[i, 1])].tolist():
            bad.append(i)
    assert not bad, f"{len(bad)} global tasks differ, first {[(len(qs[i]), len(ts[i]), int(prm[i])) for i in bad[:5]]}"
