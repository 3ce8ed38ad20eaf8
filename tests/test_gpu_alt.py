"""ALT contigs (<prefix>.alt, bwa's bntann1_t.is_alt): the index loader reads the file, a kept
chain on an ALT contig does not shadow chains on primary contigs in the chain filter (mem_chain_flt), regions and candidates
carry is_alt (the flag the reference unpacks at src/bwabridge.c:371).  Whole path against the oracle, whose loader reads the same
file; the same reads against the same index WITHOUT the file give different candidate lists, so the file is what is tested."""
import os
import shutil
import tempfile

import numpy as np
import pytest

import oracle_lib as O
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine
from test_gpu_pipeline import compare

pytestmark = pytest.mark.gpu


def _pairs(ctg):
    # reads from everywhere, and a second helping from the two ALT contigs and the primary segments they copy
    a = synth.make_pairs(ctg, 500, seed=71, sub_rate=0.01)
    b = synth.make_pairs([ctg[0][38000:66000], ctg[1][28000:44000], ctg[2], ctg[3]], 700, seed=72, sub_rate=0.004)
    bases = np.concatenate([a.bases, b.bases])
    off = np.concatenate([a.off, b.off[1:] + a.off[-1]]).astype(np.uint32)
    return synth.Pairs(bases, off)


def test_alt_contigs_whole_path_against_the_oracle():
    prefix, ctg = small_ref("with_alt")
    pairs = _pairs(ctg)
    eng = Engine(prefix)
    assert [eng._L.ema_engine_contig_is_alt(eng._h, i) for i in range(4)] == [0, 0, 1, 1]
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert batch.status.max() == 0
    bad = compare(prefix, pairs, batch)
    assert not bad, f"{len(bad)} of {2 * pairs.n} reads differ from the oracle, first {bad[:5]}"
    n_alt = int((batch.cand["is_alt"] != 0).sum())
    assert n_alt > 300 and all(int(c["rid"]) >= 2 for c in batch.cand[batch.cand["is_alt"] != 0])
    assert all(int(c["rid"]) < 2 for c in batch.cand[batch.cand["is_alt"] == 0])

    # the same index without the .alt file: no flags, and other candidate lists (ALT chains shadow primary ones again)
    d = tempfile.mkdtemp(prefix="ema_noalt_")
    for ext in ("", ".fai", ".bwt", ".fsa", ".sa", ".pac", ".ann", ".amb"):
        os.symlink(prefix + ext, os.path.join(d, "ref.fa" + ext))
    plain = os.path.join(d, "ref.fa")
    eng = Engine(plain)
    b2 = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert int((b2.cand["is_alt"] != 0).sum()) == 0
    assert not compare(plain, pairs, b2)
    assert len(b2.cand) != len(batch.cand) or not np.array_equal(b2.cand["rb"], batch.cand["rb"])
    shutil.rmtree(d)


def test_alt_contigs_lane_and_wave_routes_agree(tuning):
    """K2a's per-lane chain filter and K2b's wave-wide ones (slab and medium layout) all read the flag: the same batch with the
    lane kernels switched off."""
    prefix, ctg = small_ref("with_alt")
    pairs = _pairs(ctg)
    tuning(lane_align=0)
    eng = Engine(prefix)
    batch = eng.align_pairs(pairs.bases, pairs.off)
    eng.close()
    assert not compare(prefix, pairs, batch)
