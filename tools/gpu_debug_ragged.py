import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import small_ref
from ema_amd import synth
from ema_amd.engine import Engine, default_opts
prefix, ctg = small_ref("two_contigs")
o = default_opts(); o.batch_pairs = 64
eng = Engine(prefix, opts=o)
which = sys.argv[1]
if which == "empty":
    b = eng.align_pairs(np.zeros(0, np.uint8), np.zeros(1, np.uint32))
    print("empty ok", len(b.cand), flush=True)
else:
    pairs = synth.make_pairs(ctg, 8, seed=45)
    reads = [pairs.read(i) for i in range(16)]
    mods = {"short": (1, lambda r: r[:10]), "one": (2, lambda r: b"A"), "alln": (5, lambda r: b"N" * 60), "n19": (6, lambda r: r[:19]), "zero": (9, lambda r: b"")}
    for k in which.split(","):
        if k in mods:
            i, f = mods[k]; reads[i] = f(reads[i])
    off = np.zeros(17, np.uint32); off[1:] = np.cumsum([len(r) for r in reads])
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    b = eng.align_pairs(bases, off)
    print(which, "ok", len(b.cand), flush=True)
