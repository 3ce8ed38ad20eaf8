"""Dev aid (CPU): rate of ema_count_fastq against the reference's own `ema count` ($TMPDIR/ema_ref/ref_count, where it exists) on a
synthetic interleaved FASTQ of N pairs (2x150 bp, 10x barcodes from a 100 K whitelist), and that the four files are equal.
  python tools/count_rate.py [N_PAIRS]"""
import os, random, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from ema_amd import count as ema_count
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rng = np.random.default_rng(7)
d = tempfile.mkdtemp(prefix="ema_count_")
wl = rng.integers(0, 4, (100000, 16), dtype=np.uint8); wl[(wl == 0).all(axis=1), 0] = 1
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
open(os.path.join(d, "wl.txt"), "wb").write(b"\n".join(lut[w].tobytes() for w in wl) + b"\n")
# fixed-width records: "@rNNNNNNNN 1\n" + 150 bases + "\n+\n" + 150 quals + "\n" twice
L = 150
name = np.char.add(np.char.add("@r", np.char.zfill(np.arange(n).astype(str), 9)), " 1").astype("S").view(np.uint8).reshape(n, -1)
w = name.shape[1]
rec = np.empty((n, 2 * (w + 1 + L + 3 + L + 1)), dtype=np.uint8)
half = w + 1 + L + 3 + L + 1
bases = lut[rng.integers(0, 4, (n, 2 * L), dtype=np.uint8)]
pick = rng.integers(0, len(wl), n); bases[:, :16] = lut[wl[pick]]
off = rng.random(n) < 0.1; p = rng.integers(0, 16, n); bases[off, p[off]] = ord("N")      # 10 % with an N in the barcode
for m in range(2):
    c = m * half
    rec[:, c:c + w] = name; c += w
    rec[:, c] = 10; c += 1
    rec[:, c:c + L] = bases[:, m * L:(m + 1) * L]; c += L
    rec[:, c] = 10; rec[:, c + 1] = ord("+"); rec[:, c + 2] = 10; c += 3
    rec[:, c:c + L] = rng.choice(np.frombuffer(b"#5AFJ", dtype=np.uint8), (n, L)); c += L
    rec[:, c] = 10
fq = os.path.join(d, "in.fastq")
rec.tofile(fq)
size = os.path.getsize(fq)
t = time.perf_counter()
st = ema_count.count_fastq(os.path.join(d, "wl.txt"), fq, os.path.join(d, "a"))
dt = time.perf_counter() - t
print(f"ema_count_fastq: {n / dt:,.0f} pairs/s, {size / dt / 1e6:,.0f} MB/s ({dt:.2f} s; {st})")
ref = os.path.join(os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref"), "ref_count")
if os.path.exists(ref):
    t = time.perf_counter()
    subprocess.run([ref, os.path.join(d, "wl.txt"), os.path.join(d, "b"), str(1 << 30), "0"], stdin=open(fq, "rb"), check=True, stderr=subprocess.DEVNULL)
    dr = time.perf_counter() - t
    same = all(open(os.path.join(d, "a." + e), "rb").read() == open(os.path.join(d, "b." + e), "rb").read() for e in ("ema-fcnt", "ema-ncnt"))
    print(f"reference `ema count`: {n / dr:,.0f} pairs/s, {size / dr / 1e6:,.0f} MB/s ({dr:.2f} s); files identical: {same}")
