"""Quick K1 timing on the GPU box (development aid)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine, default_opts

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
t = time.time(); prefix, ctg = small_ref("mid"); print("index", time.time() - t, flush=True)
t = time.time(); pairs = synth.make_pairs(ctg, n_pairs, seed=9); print("reads", time.time() - t, flush=True)
o = default_opts(); o.batch_pairs = n_pairs
eng = Engine(prefix, opts=o)
eng.stage(pairs.bases, pairs.off)
for it in range(4):
    eng.run(); eng.sync()
    tm = eng.timing()
    print("seed_ms %.3f  -> %.2f M reads/s" % (tm["seed_ms"], 2 * n_pairs / tm["seed_ms"] / 1e3), flush=True)
