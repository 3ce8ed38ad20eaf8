"""Dev aid (no GPU): K1 through the host SIMT interpreter vs the oracle's mem_collect_intv."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import common, emu_lib, oracle_lib as O
from ema_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
tot_bad = 0
for kind, kw in (("repeats", {}), ("ngaps", dict(n_rate=0.01)), ("two_contigs", dict(sub_rate=0.05))):
    prefix, ctg = common.small_ref(kind)
    pairs = synth.make_pairs(ctg, n, seed=9, **kw)
    nt4 = np.array([{65: 0, 67: 1, 71: 2, 84: 3}.get(c, 4) for c in pairs.bases], dtype=np.uint8)
    off = pairs.off.astype(np.uint32)
    h = emu_lib.index_load(prefix)
    intv, n_intv, status = emu_lib.seed(h, nt4, off, n_blocks=1, wave=os.environ.get("EMU_SEED_WAVE") == "1")
    idx, opt = O.Index(prefix), O.default_opt()
    if os.environ.get("EMU_MIN_SEED_LEN"):
        opt.min_seed_len = int(os.environ["EMU_MIN_SEED_LEN"])
    bad = 0
    for r in range(len(off) - 1):
        ref = O.collect_intv(idx, opt, pairs.read(r))
        got = [(int(a[3]) >> 32, int(a[3]) & 0xffffffff, int(a[0]), int(a[1]), int(a[2])) for a in intv[r, :n_intv[r]]]
        exp = [tuple(int(t) for t in d) for d in ref]
        if not common.same_intervals(got, exp, idx, os.environ.get("EMU_KMER_K", "0") != "0"):
            bad += 1
            if bad < 3: print(kind, r, len(got), len(exp), got[:3], exp[:3])
    print(kind, "reads", len(off) - 1, "mismatching", bad, "status", np.unique(status))
    tot_bad += bad
sys.exit(1 if tot_bad else 0)
