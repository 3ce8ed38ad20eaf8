"""Dev aid (CPU, oracle): how many of a bench batch's extensions (ksw_extend2 calls of mem_chain2aln) the engine's lane-per-read pass
(K2a, lane_extend_diag) can decide without the dynamic program, why not the others, and how many READS therefore go on to the
wave-per-read kernel (K2b's hand-overs) -- the oracle replays the rule on its own calls (oracle/extend.c, orc_extprof).
  python tools/cpu_extend_profile.py [PREFIX] [READS.npz] [N_PAIRS]     (defaults: the bench workdir's reference and first batch)"""
import ctypes as C, glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from ema_amd import synth
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
prefix = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(wd, "ref.fa")
reads = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
z = np.load(reads)
pairs = synth.Pairs(z["bases"], z["off"]).subset(0, n)
idx, opt = O.Index(prefix), O.default_opt()
L = O.lib()
L.orc_extprof_enable(1)
L.orc_extprof_reset()
O.bench_pairs(idx, opt, pairs.bases, pairs.off, 1)
buf = (C.c_uint64 * 16)()
L.orc_extprof_get(buf)
h = [int(x) for x in buf]
print(f"{h[6]} mem_align1_core calls (reads, mates re-aligned for rescue excluded: none on this path), {h[0]} extensions = {h[0] / max(1, h[6]):.2f} per read")
print(f"  decided by the diagonal rule        {h[1]:9d}  {100.0 * h[1] / max(1, h[0]):5.1f} %")
for k, name in ((2, "target shorter than query / h0 <= 0"), (3, "an ambiguous base"), (4, "two or more mismatches"), (5, "one mismatch, conditions fail")):
    print(f"  not decided: {name:36s} {h[k]:9d}  {100.0 * h[k] / max(1, h[0]):5.1f} %")
print(f"  reads with an extension the rule does not decide: {h[7]} = {100.0 * h[7] / max(1, h[6]):.1f} % of reads; the first one was in the read's best chain for {h[8]} of them ({100.0 * h[8] / max(1, h[7]):.1f} %)")
