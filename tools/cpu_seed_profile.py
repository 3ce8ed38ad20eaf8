"""Dev aid (CPU, oracle): where the seeding stage's rank queries go -- bwt_extend calls by pass/direction and by the length of
the string they produce -- on a sample of a bench batch.  Tells what a k-mer interval table of depth k would take off K1.
  python tools/cpu_seed_profile.py [PREFIX] [READS.npz] [N_PAIRS]     (defaults: the bench workdir's reference and first batch)"""
import ctypes as C, glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from ema_amd import synth
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
prefix = sys.argv[1] if len(sys.argv) > 1 else os.path.join(wd, "ref.fa")
reads = sys.argv[2] if len(sys.argv) > 2 else sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
z = np.load(reads)
pairs = synth.Pairs(z["bases"], z["off"]).subset(0, n)
idx, opt = O.Index(prefix), O.default_opt()
L = O.lib()
L.orc_seedprof_reset()
O.stats_reset()
O.bench_pairs(idx, opt, pairs.bases, pairs.off, 1)
st = O.stats_get()
buf = (C.c_uint64 * (5 * 64))()
L.orc_seedprof_get(buf)
h = np.array(buf[:], dtype=np.int64).reshape(5, 64)
tot = h.sum()
names = ["pass1 forward", "pass1 backward", "pass2 forward", "pass2 backward", "pass3"]
print(f"{2 * n} reads, {tot} extends = {tot / (2 * n):.1f} per read (stats n_ext {st['n_ext']})")
for c in range(5):
    print(f"  {names[c]:15s} {h[c].sum() / (2 * n):8.1f} per read  {100.0 * h[c].sum() / tot:5.1f} %")
for k in (8, 10, 11, 12, 13, 14, 15, 16, 18, 20):
    cover = h[:, :k + 1].sum()
    per = [100.0 * h[c, :k + 1].sum() / max(1, h[c].sum()) for c in range(5)]
    print(f"  result length <= {k:2d}: {100.0 * cover / tot:5.1f} % of all extends   by class: " + " ".join(f"{x:5.1f}" for x in per))
print("  histogram of result lengths (all classes):", h.sum(axis=0).tolist())
