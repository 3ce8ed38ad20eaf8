"""Development aid: cost per DP row of the three wave DPs (one task per wavefront)."""
import sys, os
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
n = 200000
q = rng.integers(0, 4, 75).astype(np.uint8)
t = np.concatenate([q, rng.integers(0, 4, 35).astype(np.uint8)])
for kind, prm in ((0, [100, 5, 100, 50]), (1, [20]), (2, [16, 19, 0x10000])):
    qs = [q] * n; ts = [t if kind != 1 else q] * n
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    p = np.tile(np.array(prm, np.int32), (n, 1))
    for _ in range(2):
        out, _c = eng.debug_dp(kind, qb, qo, tb, to, p)
    print(kind, out[0].tolist(), flush=True)
