"""Dev aid (GPU + CPU): how well k-mer counts predict the reads that exceed the lean tier's seeding budget (and are seeded a second
time, from scratch, by the full tier's wave-per-read kernel).  One bench batch: the lean status bits of every read (full tier made
tiny, as tools/gpu_capdist.py does), the occurrence counts of all 12-mers of the bench genome (both strands, numpy), and for each read
the counts of the six 12-mers ema_k_seed_order samples; then precision / recall of a few statistics and thresholds.
  python tools/gpu_long_read_predictor.py [N_READS]"""
import glob, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd.engine import Engine, default_opts
K = 12
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
bases, off = z["bases"], z["off"]
o = default_opts(); o.batch_pairs = (len(off) - 1) // 2
o.full_tier_pairs = 64
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
eng.stage(bases, off)
eng.run(); eng.sync()
b = eng.fetch(allow_limit=True)
long_read = (b.status & 256) != 0
eng.close()
print(f"{len(long_read)} reads, {int(long_read.sum())} over the lean seeding budget ({100.0 * long_read.mean():.3f} %)", flush=True)

t = time.time()
g = np.load(os.path.join(wd, "genome.npy"), mmap_mode="r")
cnt = np.zeros(1 << (2 * K), dtype=np.uint32)
CH = 1 << 26
mask = (1 << (2 * K)) - 1
for a in range(0, len(g), CH):
    x = np.asarray(g[a:min(len(g), a + CH + K - 1)]).astype(np.uint32)
    ok = x < 4
    code = np.zeros(len(x) - K + 1, dtype=np.uint32)
    good = np.ones(len(x) - K + 1, dtype=bool)
    for j in range(K):
        code = (code << 2) | (x[j:len(x) - K + 1 + j] & 3)
        good &= ok[j:len(x) - K + 1 + j]
    cnt += np.bincount(code[good], minlength=1 << (2 * K)).astype(np.uint32)
# the other strand: count of a k-mer's reverse complement
idx = np.arange(1 << (2 * K), dtype=np.uint32)
rc = np.zeros_like(idx)
for j in range(K):
    rc = (rc << 2) | (3 - ((idx >> (2 * j)) & 3))
both = cnt + cnt[rc]
print(f"12-mer counts of {len(g)} bases x 2 strands in {time.time() - t:.0f} s; mean {both.mean():.1f}", flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else len(long_read)
lut = np.full(256, 4, np.uint8)
for i, c in enumerate(b"ACGT"):
    lut[c] = i
Ls = (off[1:n + 1] - off[:n]).astype(np.int64)
stats = np.zeros((n, 6), dtype=np.uint32)
pw = (4 ** np.arange(K - 1, -1, -1)).astype(np.int64)
for j in range(6):
    p = (Ls - K) * j // 5
    okl = Ls >= K
    at = off[:n].astype(np.int64) + np.where(okl, p, 0)
    w = lut[bases[(at[:, None] + np.arange(K)[None, :]).clip(0, len(bases) - 1)]]
    good = okl & (w < 4).all(axis=1)
    code = ((w & 3).astype(np.int64) * pw[None, :]).sum(axis=1)
    stats[:, j] = np.where(good, both[code], 0)
expected = 2 * len(g) / 4 ** K
lr = long_read[:n]
srt = np.sort(stats, axis=1)
print(f"expected count of a random 12-mer {expected:.0f}")
for name, v in (("max", srt[:, 5]), ("2nd largest", srt[:, 4]), ("median (3rd largest)", srt[:, 3]), ("4th largest", srt[:, 2]), ("min", srt[:, 0]), ("sum / 6", stats.sum(axis=1) / 6)):
    print(f"statistic: {name}")
    for mult in (4, 8, 16, 32, 64, 128, 256):
        pred = v > mult * expected
        tp = int((pred & lr).sum())
        print(f"   > {mult:4d} x expected: {int(pred.sum()):8d} reads ({100.0 * pred.mean():6.2f} %), of them over the budget {tp:7d} = precision {100.0 * tp / max(1, pred.sum()):5.1f} %, recall {100.0 * tp / max(1, lr.sum()):5.1f} %")
