"""A stock bwa index (reference src/bwabridge.c:77-96: bwa_idx_load reads .bwt .sa .pac .ann .amb and nothing
else): without this repo's flat suffix array (<prefix>.fsa) the engine expands bwa's sampled .sa on the device at open
(k_kmer.hip, ema_k_sa_expand: bwt_sa()'s LF walk for every row).  The expanded array equals the builder's row for row, and the whole
path gives the oracle's candidates on it -- 4-byte and 8-byte rows, one and several rank superblocks' worth of index."""
import os
import shutil

import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine

pytestmark = pytest.mark.gpu


def _stock_copy(prefix, tmp_path):
    """The index files `bwa index` writes (and .alt when there is one), without the .fsa"""
    dst = str(tmp_path / "stock.fa")
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb", ".alt"):
        if os.path.exists(prefix + ext):
            shutil.copy(prefix + ext, dst + ext)
    assert not os.path.exists(dst + ".fsa")
    return dst


@pytest.mark.parametrize("sa64", ["0", "1"])
@pytest.mark.parametrize("kind", ["repeats", "with_alt"])
def test_engine_on_a_stock_bwa_index(kind, sa64, tmp_path, monkeypatch):
    monkeypatch.setenv("EMA_INDEX_SA64", sa64)      # 1: 8-byte rows, as a human-size index gets
    prefix, ctg = small_ref(kind)
    stock = _stock_copy(prefix, tmp_path)
    eng = Engine(stock)
    n_rows = 2 * sum(len(c) for c in ctg) + 1
    got = eng.debug_sa(0, n_rows)
    head = np.fromfile(prefix + ".fsa", dtype=np.uint64, count=3)
    width = int(head[2])
    want = np.fromfile(prefix + ".fsa", dtype=np.uint32 if width == 4 else np.uint64, offset=24).astype(np.uint64)
    assert len(want) == n_rows and (got == want).all(), f"{int((got != want).sum())} suffix-array rows differ"
    pairs = synth.make_pairs(ctg, 300, seed=61, sub_rate=0.01)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    idx, opt = O.Index(prefix), O.default_opt()
    bad = 0
    for p in range(pairs.n):
        ref = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            g = [(int(c["rb"]), int(c["re"]), int(c["qb"]), int(c["qe"]), int(c["score"]), int(c["pos"]), int(c["is_rev"]), int(c["NM"]),
                  int(c["is_alt"]), batch.cigar_of(c).tolist()) for c in batch.mate(p, m)]
            e = [(d["rb"], d["re"], d["qb"], d["qe"], d["score"], d["pos"], d["is_rev"], d["NM"], d["is_alt"], d["cigar"]) for d in ref[m]]
            bad += g != e
    assert bad == 0, f"{bad} reads differ from the oracle on the stock index"
