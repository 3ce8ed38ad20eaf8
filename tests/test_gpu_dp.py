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
