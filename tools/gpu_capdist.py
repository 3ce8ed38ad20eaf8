"""Dev aid: per-read distributions of seed intervals, candidates and CIGAR ops on the bench workload (sizes the lean tier)."""
import os, sys, argparse, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, __graft_entry__
__graft_entry__.ensure_built()
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=65536)
ap.add_argument("--genome-mbp", type=float, default=0.0)
a = ap.parse_args()
workdir = os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid()); os.makedirs(workdir, exist_ok=True)
prefix, pairs, _ = bench.build_workload(a, 0, 1, workdir)
from ema_amd.engine import Engine, default_opts
o = default_opts(); o.batch_pairs = a.pairs; o.n_streams = 1
eng = Engine(prefix, device=0, opts=o)
eng.stage(pairs.bases, pairs.off)
intv, n_intv = eng.debug_seeds()
def dist(name, v):
    v = np.asarray(v)
    qs = [50, 90, 99, 99.9, 99.99, 100]
    print(name, "mean %.1f" % v.mean(), " ".join("p%g=%d" % (q, np.percentile(v, q)) for q in qs), flush=True)
    for cap in (16, 24, 32, 48, 64, 96, 128, 192, 256):
        print("   >%d: %.5f%%" % (cap, 100.0 * (v > cap).mean()))
dist("n_intv", n_intv)
eng.run(); eng.sync()
b = eng.fetch(allow_limit=True)
nc = np.diff(b.cand_off)
dist("n_cand", nc)
cig = np.zeros(len(nc), dtype=np.int64)
np.add.at(cig, np.repeat(np.arange(len(nc)), nc), b.cand["n_cigar"].astype(np.int64))
dist("cigar_ops", cig)
print("status flags:", np.unique(b.status, return_counts=True))
