"""Dev aid (no GPU): K1..K4 through the host SIMT interpreter vs the oracle (candidates, positions, NM, CIGARs)."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import common, emu_lib, oracle_lib as O
from ema_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tot = 0
for kind, kw in (("two_contigs", dict(sub_rate=0.02, indel_rate=0.004)), ("repeats", {}), ("ngaps", dict(n_rate=0.01))):
    prefix, ctg = common.small_ref(kind)
    pairs = synth.make_pairs(ctg, n, seed=13, **kw)
    nt4 = np.array([{65: 0, 67: 1, 71: 2, 84: 3}.get(c, 4) for c in pairs.bases], dtype=np.uint8)
    off = pairs.off.astype(np.uint32)
    h = emu_lib.index_load(prefix)
    regs, n_regs, alns, cigars, cig_n, status = emu_lib.pipeline(h, nt4, off)
    idx, opt = O.Index(prefix), O.default_opt()
    bad = 0
    for p in range(pairs.n):
        ref = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
        for m in range(2):
            r = 2 * p + m
            got = [(int(regs[r, k]["rb"]), int(regs[r, k]["re"]), int(regs[r, k]["qb"]), int(regs[r, k]["qe"]), int(regs[r, k]["score"]),
                    int(alns[r, k]["pos"]), int(alns[r, k]["is_rev"]), int(alns[r, k]["NM"]),
                    cigars[r, alns[r, k]["cigar_off"]:alns[r, k]["cigar_off"] + alns[r, k]["n_cigar"]].tolist()) for k in range(n_regs[r])]
            exp = [(d["rb"], d["re"], d["qb"], d["qe"], d["score"], d["pos"], d["is_rev"], d["NM"], d["cigar"]) for d in ref[m]]
            if got != exp:
                bad += 1
                if bad < 3: print(kind, r, got[:2], exp[:2])
    print(kind, "reads", 2 * pairs.n, "mismatching", bad, "status", np.unique(status))
    tot += bad
sys.exit(1 if tot else 0)
