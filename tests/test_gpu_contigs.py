"""The device's contig look-ups against bwa's (bns_pos2rid / bns_intv2rid of the un-vendored bwa, reached from mem_chain at every
seed occurrence and from bns_fetch_seq at every window; reference src/bwabridge.c:236-237).  The engine answers them from a
coarse table and a bisection of what is left (ema_amd/csrc/dev_ref.hpp); here on layouts where many contigs start inside one
block of the table, which no test genome of the suite has."""
import numpy as np
import pytest

from common import small_ref
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu


def bwa_pos2rid(off, pos_f):
    """bns_pos2rid: bisection over the contig offsets (-1 beyond l_pac)."""
    l_pac, n = off[-1], len(off) - 1
    if pos_f >= l_pac:
        return -1
    left, mid, right = 0, 0, n
    while left < right:
        mid = (left + right) >> 1
        if pos_f >= off[mid]:
            if mid == n - 1 or pos_f < off[mid + 1]:
                break
            left = mid + 1
        else:
            right = mid
    return mid


def bwa_intv2rid(off, rb, re):
    l_pac = off[-1]
    if rb < l_pac and re > l_pac:
        return -2
    depos = lambda p: (2 * l_pac - 1 - p) if p >= l_pac else p
    b = bwa_pos2rid(off, depos(rb))
    e = bwa_pos2rid(off, depos(re - 1)) if rb < re else b
    return b if b == e else -1


@pytest.fixture(scope="module")
def eng():
    prefix, _ = small_ref("two_contigs")      # any index: the look-ups run on the layout the test passes in
    e = Engine(prefix)
    yield e
    e.close()


@pytest.mark.parametrize("layout", ["tiny_contigs_in_a_large_reference", "one_contig", "small_reference", "human_like"])
def test_contig_lookups_equal_bwas(eng, layout):
    rng = np.random.default_rng(11)
    if layout == "tiny_contigs_in_a_large_reference":      # 2^31 bases: blocks of 2^16; thousands of 200-3000 base contigs between long ones
        lens = []
        for _ in range(40):
            lens.append(int(rng.integers(20_000_000, 60_000_000)))
            lens += [int(x) for x in rng.integers(200, 3000, int(rng.integers(1, 120)))]
    elif layout == "one_contig":
        lens = [5_000_000]
    elif layout == "small_reference":                      # below 2^16 bases: one table entry per base
        lens = [int(x) for x in rng.integers(50, 900, 60)]
    else:                                                  # 24 chromosomes and 3000 short decoys at the end
        lens = [int(x) for x in rng.integers(40_000_000, 250_000_000, 24)] + [int(x) for x in rng.integers(500, 40_000, 3000)]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    l_pac = int(off[-1])
    n = 20000
    # queries: around contig starts and ends (both strands), random ones, the strand junction, the far ends
    edges = rng.choice(off, n // 2)
    rb = edges + rng.integers(-40, 40, n // 2)
    rb = np.concatenate([rb, rng.integers(0, 2 * l_pac, n // 2 - 8), [0, l_pac - 1, l_pac, 2 * l_pac - 1, l_pac - 10, l_pac - 1, 1, l_pac + 1]])
    rev = rng.random(n) < 0.5
    rb = np.where(rev, 2 * l_pac - 1 - rb, rb)
    rb = np.clip(rb, 0, 2 * l_pac - 1).astype(np.int64)
    ln = rng.integers(0, 160, n)
    ln[rng.random(n) < 0.05] = 0
    re = np.minimum(rb + ln, 2 * l_pac).astype(np.int64)
    got_intv, got_pos = eng.debug_contigs(off, rb, re)
    offl = [int(x) for x in off]
    depos = lambda p: (2 * l_pac - 1 - p) if p >= l_pac else p
    want_intv = np.array([bwa_intv2rid(offl, int(b), int(e)) for b, e in zip(rb, re)], dtype=np.int32)
    want_pos = np.array([bwa_pos2rid(offl, depos(int(b))) for b in rb], dtype=np.int32)
    assert np.array_equal(got_pos, want_pos), np.flatnonzero(got_pos != want_pos)[:10]
    assert np.array_equal(got_intv, want_intv), np.flatnonzero(got_intv != want_intv)[:10]
    assert (want_intv >= 0).sum() > 0 and (len(lens) == 1 or (want_intv == -1).sum() > 0)


def test_tuning_string_takes_lists_with_colons(tuning):
    """`grid` and `seed_order` are lists inside a comma-separated tuning string (with commas inside the value only the
    first number arrived).  The wrapper turns a legacy "a,b,c,d" into "a:b:c:d"; all four grids must follow."""
    prefix, _ = small_ref("two_contigs")
    e = Engine(prefix)
    free = e.debug_grids()
    e.close()
    assert min(free[k] for k in ("k2a", "k2b", "k3", "k4")) >= 2, free
    tuning(grid="1,1,1,1", seed_blocks_per_cu=1)      # (the wrapper turns the commas of a list into colons)
    e = Engine(prefix)
    held = e.debug_grids()
    e.close()
    assert [held[k] for k in ("k1", "k2a", "k2b", "k3", "k4")] == [1, 1, 1, 1, 1], held
    tuning(grid="0:1:0:1")
    e = Engine(prefix)
    mixed = e.debug_grids()
    e.close()
    assert (mixed["k2a"], mixed["k2b"], mixed["k3"], mixed["k4"]) == (free["k2a"], 1, free["k3"], 1), mixed
