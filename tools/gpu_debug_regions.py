"""Development aid: run K1+K2 on a handful of pairs with progress prints (used under `timeout`)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine, default_opts
import oracle_lib as O

kind = sys.argv[1]; n = int(sys.argv[2]); upto = sys.argv[3] if len(sys.argv) > 3 else "regions"
prefix, ctg = small_ref(kind)
pairs = synth.make_pairs(ctg, n, seed=2)
o = default_opts(); o.batch_pairs = max(n, 64)
eng = Engine(prefix, opts=o)
print("engine open", flush=True)
eng.stage(pairs.bases, pairs.off)
t = time.time()
if upto == "regions":
    regs, n_regs, status = eng.debug_regions()
    print("regions done %.3fs" % (time.time() - t), "status", status.max(), "mean regs", n_regs.mean(), flush=True)
else:
    eng.run(); eng.sync()
    print("run done %.3fs" % (time.time() - t), eng.timing(), flush=True)
    b = eng.fetch()
    print("fetch done", len(b.cand), flush=True)
