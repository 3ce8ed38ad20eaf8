import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import small_ref
from ema_amd.engine import Engine, default_opts, REG_DTYPE
prefix, ctg = small_ref("two_contigs")
o = default_opts(); o.batch_pairs = 64
eng = Engine(prefix, opts=o)
rows = [(332715, 332839, 3, 127, 64), (418240, 418261, 97, 118, 21), (595232, 595262, 97, 127, 21)]
regs = np.zeros((1, 8), dtype=REG_DTYPE)
for i, (rb, re, qb, qe, sc) in enumerate(rows):
    regs[0, i]["rb"] = rb; regs[0, i]["re"] = re; regs[0, i]["qb"] = qb; regs[0, i]["qe"] = qe; regs[0, i]["score"] = sc
print("calling", flush=True)
out, n = eng.debug_dedup(regs, np.array([3], np.int32))
print("n_out", n, [(int(x["rb"]), int(x["score"])) for x in out[0, :n[0]]], flush=True)
