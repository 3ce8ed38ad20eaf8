"""Dev aid (no GPU): the two capacity tiers through the host SIMT interpreter vs the single-tier pipeline."""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import common, emu_lib
from ema_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
prefix, ctg = common.small_ref("repeats")
pairs = synth.make_pairs(ctg, n, seed=5)
nt4 = np.array([{65: 0, 67: 1, 71: 2, 84: 3}.get(c, 4) for c in pairs.bases], dtype=np.uint8)
off = pairs.off.astype(np.uint32)
h = emu_lib.index_load(prefix)
ref = emu_lib.pipeline(h, nt4, off)
got = emu_lib.pipeline_tiers(h, nt4, off, lean=(8, 2, 8), full_pairs=max(4, n))
regs, n_regs, alns, cigars, cig_n, status = ref
g_regs, g_n, g_alns, g_cig, g_cn, g_st, tier, listed = got
print("listed pairs", listed, "tiers", np.unique(tier, return_counts=True), "status", np.unique(g_st, return_counts=True))
bad = 0
for r in range(len(off) - 1):
    ok = n_regs[r] == g_n[r] and cig_n[r] == g_cn[r] and (regs[r, :n_regs[r]] == g_regs[r, :n_regs[r]]).all() \
        and (alns[r, :n_regs[r]] == g_alns[r, :n_regs[r]]).all() and (cigars[r, :cig_n[r]] == g_cig[r, :cig_n[r]]).all()
    if not ok:
        bad += 1
        print("read", r, "tier", tier[r], "n_regs", n_regs[r], g_n[r], "cig", cig_n[r], g_cn[r], "status", status[r], g_st[r])
print("mismatching reads:", bad)
