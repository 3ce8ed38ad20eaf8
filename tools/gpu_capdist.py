"""Dev aid (GPU): which lean-tier capacity sends pairs to the full-capacity tier, on one bench batch at the bench's scale.  The full
tier is made tiny, so that the pairs it cannot take keep their lean status bits, and the bits are counted (1 intervals, 2 lists,
4 seeds, 8 chains, 16 regions, 32 reference window, 64 CIGAR operations, 256 seeding budget).
  python tools/gpu_capdist.py [lean_intervals lean_regions lean_cigar_ops [lean_seed_extends]]"""
import glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd.engine import Engine, default_opts
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
o = default_opts(); o.batch_pairs = (len(z["off"]) - 1) // 2
o.full_tier_pairs = 64
a = [int(x) for x in sys.argv[1:]]
if len(a) >= 3:
    o.lean_intervals, o.lean_regions, o.lean_cigar_ops = a[:3]
if len(a) >= 4:
    o.lean_seed_extends = a[3]
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
eng.stage(z["bases"], z["off"])
eng.run(); eng.sync()
b = eng.fetch(allow_limit=True)
st = b.status & ~128
n_pairs = len(st) // 2
pair = st[0::2] | st[1::2]
print(f"lean capacities {a[:3] if len(a) >= 3 else 'default (48 intervals, 48 regions, 192 CIGAR operations)'}; {n_pairs} pairs, {int((pair != 0).sum())} with a lean flag "
      f"({100.0 * (pair != 0).mean():.3f} %)")
for bit, name in ((1, "intervals"), (2, "lists"), (4, "seeds"), (8, "chains"), (16, "regions"), (32, "reference window"), (64, "CIGAR operations"), (256, "seeding budget")):
    n = int(((pair & bit) != 0).sum()); only = int((pair == bit).sum())
    if n:
        print(f"  {name:18s}: {n:7d} pairs ({100.0 * n / n_pairs:.3f} %), {only} of them with no other flag")
eng.close()
