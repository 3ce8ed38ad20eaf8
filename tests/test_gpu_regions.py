"""GPU parity of K1+K2 (seeding -> chaining -> extension -> dedup, bwa's mem_align1_core) against the oracle."""
import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu
FIELDS = [f for f in O.REG_FIELDS if f != "sub_n"]


def _check(kind, n_pairs, seed, **kw):
    prefix, ctg = small_ref(kind)
    pairs = synth.make_pairs(ctg, n_pairs, seed=seed, **kw)
    eng = Engine(prefix)
    eng.stage(pairs.bases, pairs.off)
    regs, n_regs, status = eng.debug_regions()
    eng.close()
    assert status.max() == 0, "a read exceeded an engine capacity"
    idx, opt = O.Index(prefix), O.default_opt()
    bad = []
    for r in range(2 * pairs.n):
        ref = O.align1(idx, opt, pairs.read(r))
        got = [{f: (float(x[f]) if f == "frac_rep" else int(x[f])) for f in FIELDS} for x in regs[r, :n_regs[r]]]
        for d in ref:
            d["frac_rep"] = float(np.float32(d["frac_rep"]))
            d.pop("sub_n")
        if ref != got:
            bad.append(r)
    assert not bad, f"{len(bad)} of {2 * pairs.n} reads have different regions, first {bad[:5]}"


def test_regions_clean():
    _check("two_contigs", 1000, 31)


def test_regions_with_n_and_indels():
    _check("two_contigs", 600, 32, n_rate=0.005, indel_rate=0.004, sub_rate=0.02)


def test_regions_repeats():
    _check("repeats", 1500, 33)


def test_regions_ngaps_reference():
    _check("ngaps", 500, 34)


def test_regions_250bp():
    _check("repeats", 300, 35, len1=250, len2=250, indel_rate=0.002)


def test_regions_repeat_family():
    """Reads from a 640-copy diverged repeat: dozens to several hundred seed occurrences and chains per read -- K2b's medium
    layout (chain positions, filter keys and kept list in LDS up to 256 chains) and its move to the slab beyond that."""
    _check("repeat_family", 250, 36)
