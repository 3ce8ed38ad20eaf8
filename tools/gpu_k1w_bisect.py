import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import common
from ema_amd import synth
from ema_amd.engine import Engine, default_opts
mode = sys.argv[1]
prefix, ctg = common.small_ref("two_contigs")
pairs = synth.make_pairs(ctg, int(sys.argv[2]) if len(sys.argv) > 2 else 4, seed=21)
o = default_opts()
if "p3" in mode: o.max_mem_intv = 0
if "p2" in mode: o.split_factor = 1e6
if "short" in mode:
    reads = [pairs.read(i)[:10] for i in range(2 * pairs.n)]
    off = np.zeros(len(reads) + 1, np.uint32); off[1:] = np.cumsum([len(r) for r in reads])
    pairs = synth.Pairs(np.frombuffer(b"".join(reads), dtype=np.uint8), off)
eng = Engine(prefix, opts=o)
eng.stage(pairs.bases, pairs.off)
intv, n_intv = eng.debug_seeds()
print(mode, "ok", n_intv[:8])
