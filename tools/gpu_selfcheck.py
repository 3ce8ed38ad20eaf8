"""Dev aid (GPU): one full bench batch (1 Mi pairs, GRCh38-scale reference) through the engine on its default routes and again with
the alternative routes forced (no K2a pre-pass, no setting-aside of chain-rich reads, no k-mer table, one slice), and the two
results compared candidate for candidate -- every field, every CIGAR operation, all pairs: the size-independent check beside the
oracle spot check of bench.py (the alternative routes are the ones the parity suite holds against the oracle read by read).
  python tools/gpu_selfcheck.py"""
import glob, hashlib, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd.engine import Engine, default_opts
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
bases, off = z["bases"], z["off"]


def run(env):
    for k in ("EMA_LANE_ALIGN", "EMA_HEAVY_CHAINS", "EMA_KMER_K"):
        os.environ.pop(k, None)
    os.environ.update(env)
    o = default_opts(); o.batch_pairs = (len(off) - 1) // 2
    if env:
        o.n_streams = 1
    eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
    b = eng.align_pairs(bases, off)
    eng.close()
    assert b.status.max() == 0
    return b


a = run({})
b = run({"EMA_LANE_ALIGN": "0", "EMA_HEAVY_CHAINS": "0", "EMA_KMER_K": "0"})
assert np.array_equal(a.cand_off, b.cand_off), "candidate counts differ"
fields = [f for f in a.cand.dtype.names if f != "cigar_off"]
for f in fields:
    assert np.array_equal(a.cand[f], b.cand[f]), f"field {f} differs"
# CIGARs candidate by candidate (the pools may be laid out differently)
n = a.cand["n_cigar"].astype(np.int64)
ia = np.repeat(a.cand["cigar_off"].astype(np.int64), n) + (np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n))
ib = np.repeat(b.cand["cigar_off"].astype(np.int64), n) + (np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n))
assert np.array_equal(a.cigar[ia], b.cigar[ib]), "CIGAR operations differ"
h = hashlib.sha256()
for f in fields:
    h.update(np.ascontiguousarray(a.cand[f]).tobytes())
h.update(np.ascontiguousarray(a.cigar[ia]).tobytes())
print(f"{(len(off) - 1) // 2} pairs, {len(a.cand)} candidates, {int(n.sum())} CIGAR operations: identical on both routes; sha256 {h.hexdigest()[:16]}; "
      f"{len(a.redone)} pairs through the full-capacity tier")
