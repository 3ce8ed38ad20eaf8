"""Index builder (ema_amd/csrc/index_build.cpp) on texts chosen to stress its parallel phases: long single-base runs and
tandem repeats (huge buckets, suffixes that tie for hundreds of bases and end inside the comparison words), contigs
shorter than the bucket key, and a reference large enough for several threads per phase, each against a suffix array
obtained independently (sorted Python slices; prefix doubling in numpy for the large one)."""
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as O
from ema_amd import build_index, synth


def build(contigs):
    d = tempfile.mkdtemp(prefix="ema_ib_")
    prefix = os.path.join(d, "r.fa")
    synth.write_fasta(prefix, contigs)
    build_index(prefix)
    text = np.concatenate(contigs)
    return prefix, np.concatenate([text, (3 - text)[::-1]]).astype(np.uint8)


def flat_sa(prefix):
    raw = np.fromfile(prefix + ".fsa", dtype=np.uint8)
    assert bytes(raw[:8]) == b"EMAFSA01"
    n, width = np.frombuffer(raw[8:24].tobytes(), dtype=np.uint64)
    sa = np.frombuffer(raw[24:].tobytes(), dtype=np.uint32 if width == 4 else np.uint64)
    assert len(sa) == n + 1
    return sa.astype(np.int64)


def doubling_sa(T):
    """Suffix array with the empty suffix first, by prefix doubling on (rank, rank at +h) pairs."""
    n = len(T)
    rank = np.concatenate([T.astype(np.int64) + 1, [0]])      # position n = '$', smallest
    sa = np.argsort(rank, kind="stable")
    h = 1
    while True:
        nxt = np.concatenate([rank[h:], np.zeros(min(h, n + 1), dtype=np.int64)])[: n + 1]
        key = rank * (n + 2) + nxt
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        new = np.zeros(n + 1, dtype=np.int64)
        new[sa] = np.concatenate([[0], np.cumsum(ks[1:] != ks[:-1])])
        rank = new
        if rank.max() == n:
            return sa
        h *= 2


CASES = {
    "one_base": [np.array([2], np.uint8)],
    "poly_a": [np.zeros(700, np.uint8)],
    "poly_a_and_t": [np.zeros(300, np.uint8), np.full(200, 3, np.uint8)],      # the reverse strand of one is the other
    "tandem_2": [np.tile(np.array([0, 1], np.uint8), 400)],
    "tandem_37": [np.tile(np.random.default_rng(1).integers(0, 4, 37).astype(np.uint8), 30)],
    "short_contigs": [np.array(x, np.uint8) for x in ([0], [1, 2], [3, 3, 3], [0, 1, 2, 3, 0, 1, 2], [2] * 9)],
    "palindromes": [np.array([0, 1, 2, 3] * 50 + [3, 2, 1, 0] * 50, np.uint8)],
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_small_adversarial_texts(name):
    prefix, T = build(CASES[name])
    s = bytes(T.tolist())
    want = sorted(range(len(T) + 1), key=lambda i: s[i:])
    assert flat_sa(prefix).tolist() == want
    idx = O.Index(prefix)      # the oracle reads .bwt/.sa: sampled rows and rank counts agree with the same array
    for row in range(1, len(T) + 1, 5):
        assert idx.sa(row) == want[row]


def test_threaded_phases_on_a_repeat_rich_reference():
    rng = np.random.default_rng(2)
    c1 = rng.integers(0, 4, 900_000).astype(np.uint8)
    c1[100_000:160_000] = c1[400_000:460_000]                      # a 60 kb exact duplication
    c1[700_000:701_000] = 0                                        # a 1 kb single-base run
    c2 = np.tile(rng.integers(0, 4, 5000).astype(np.uint8), 40)    # 200 kb of tandem copies
    prefix, T = build([c1, c2])
    assert (flat_sa(prefix) == doubling_sa(T)).all()


def test_flat_suffix_array_from_bwas_sampled_one(tmp_path):
    """host_expand_sa (host_index.cpp; the device kernel ema_k_sa_expand restates it): bwa's bwt_sa() walk over every row of a stock
    index (sampled .sa, no .fsa) gives the builder's flat suffix array -- through the host SIMT harness's loader, which keeps the
    rows in host memory."""
    import os, shutil
    import numpy as np
    import emu_lib
    from common import small_ref
    prefix, ctg = small_ref("two_contigs")
    dst = str(tmp_path / "stock.fa")
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        shutil.copy(prefix + ext, dst + ext)
    L = emu_lib.lib()
    L.emu_index_sa.restype = None
    n_rows = 2 * sum(len(c) for c in ctg) + 1
    got = np.zeros(n_rows, dtype=np.uint64)
    want = np.zeros(n_rows, dtype=np.uint64)
    for pfx, out in ((dst, got), (prefix, want)):
        h = emu_lib.index_load(pfx)
        L.emu_index_sa.argtypes = [emu_lib.C.c_void_p, emu_lib.C.c_void_p, emu_lib.C.c_uint64]
        L.emu_index_sa(h, out.ctypes.data, n_rows)
        L.emu_index_free(h)
    assert (got == want).all()
    fsa = np.fromfile(prefix + ".fsa", dtype=np.uint32, offset=24).astype(np.uint64)
    assert (want == fsa).all()
