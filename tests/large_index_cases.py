"""Parity cases for the index code paths that only a human-size reference reaches in the product build:
  * several rank superblocks (ema_lane_occ4's `n_super > 1` branch, dev_common.hpp; host side host_index.cpp), and
  * 8-byte suffix-array rows (ema_sa, dev_common.hpp).
Not collected by itself: tests/test_gpu_large_index.py runs this file in a child pytest with
  EMA_ENGINE_LIB=libema_engine_ss16.so   the same sources built with EMA_OCC_SUPER_SHIFT=16 (Makefile), and/or
  EMA_INDEX_SA64=1                       the index builder writes 8-byte rows whatever the size,
on a 120 Kbp reference (240 K BWT symbols = 4 superblocks of 2^16).  Same checks as test_gpu_seed / _regions /
_pipeline: seed intervals, regions before rescue, final candidate lists with CIGARs, all against the oracle."""
import os

import pytest

import test_gpu_pipeline as TP
import test_gpu_regions as TR
import test_gpu_seed as TS
from common import small_ref
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu
KIND = "tiny_repeats"


def test_configuration_is_what_the_parent_asked_for():
    prefix, _ = small_ref(KIND)
    eng = Engine(prefix)
    info = eng.index_info()
    eng.close()
    if os.environ.get("EMA_ENGINE_LIB", "").endswith("ss16.so"):
        assert info["n_super"] == 4 and info["super_shift"] == 16
    if os.environ.get("EMA_INDEX_SA64") == "1":
        assert info["sa_width"] == 8


@pytest.mark.parametrize("kernel", ["lane", "wave"])
def test_seeds(kernel, tuning):
    TS._check(KIND, 500, 61, kernel, tuning)


def test_seeds_with_n(tuning):
    TS._check(KIND, 300, 62, "lane", tuning, n_rate=0.01)


def test_regions():
    TR._check(KIND, 800, 63)


def test_pipeline():
    TP._check(KIND, 1000, 64)


def test_pipeline_noisy():
    TP._check(KIND, 500, 65, sub_rate=0.06, indel_rate=0.003)


def test_chain_rich_reads_on_the_large_index_paths():
    """Reads with more than EMA_MED_CHAINS = 256 chains (K2b's medium layout outgrown, the slab layout, K2c / K2d) in the same run
    as the several-superblock and 8-byte-row branches: a 118 Kbp reference with 440 copies of a 200 bp element."""
    import oracle_lib as O
    from ema_amd import synth
    prefix, ctg = small_ref("tiny_family")
    pairs = synth.make_pairs([ctg[0][2000:2000 + 260 * 440]], 160, seed=66, sub_rate=0.01)
    idx, opt = O.Index(prefix), O.default_opt()
    rich = sum(O.n_chains(idx, opt, pairs.read(r)) > 256 for r in range(0, 2 * pairs.n, 7))
    assert rich >= 5, rich
    eng = Engine(prefix)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0
    assert not TP.compare(prefix, pairs, batch)
