"""Development aid (GPU): cost per DP row of the extension DP (one task per wavefront, 64-thread blocks: the chip full of DPs and
nothing else), on extensions the diagonal does not decide -- three mismatches spread over the query, so that the row loop runs to
the query's end and the exact early exit -- for the one-, two- and three-column layouts.  EMA_DP_TIMING=1 is set here.
  python tools/gpu_dp_microbench.py [EMA_ENGINE_LIB=<variant library> in the environment]"""
import sys, os
os.environ["EMA_DP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import dp_cases as D
from common import small_ref
from ema_amd.engine import Engine, default_opts
prefix, _ = small_ref("two_contigs")
o = default_opts(); o.batch_pairs = 64
eng = Engine(prefix, opts=o)
rng = np.random.default_rng(1)
n = 400000
for qlen in (50, 100, 150):
    q = rng.integers(0, 4, qlen).astype(np.uint8)
    t = np.concatenate([q, rng.integers(0, 4, 60).astype(np.uint8)])
    for p in (qlen // 5, qlen // 2, qlen - 8):      # three mismatches
        t[p] = (t[p] + 1) & 3
    qs = [q] * n; ts = [t] * n
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    p = np.tile(np.array([100, 5, 100, 50], np.int32), (n, 1))      # w, end_bonus, zdrop, h0
    for _ in range(2):
        out, _c = eng.debug_dp(0, qb, qo, tb, to, p)
    print("qlen", qlen, "result", out[0].tolist(), flush=True)
eng.close()
