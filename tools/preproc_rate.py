"""Dev aid (CPU): rate of ema_count_fastq + ema_preproc_fastq against the reference's own `ema count` + `ema preproc`
($TMPDIR/ema_ref/ref_count, ref_preproc, where they exist) on a synthetic interleaved FASTQ of N pairs (2x150 bp, 10x barcodes from a
100 K whitelist, 10 % of the barcodes one base off, 3 % with an N), and that every bucket file is equal.
  python tools/preproc_rate.py [N_PAIRS] [N_BUCKETS]"""
import hashlib, os, subprocess, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd import count as ema_count, preproc as ema_preproc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(11)
d = tempfile.mkdtemp(prefix="ema_preproc_")
wl = rng.integers(0, 4, (100000, 16), dtype=np.uint8); wl[(wl == 0).all(axis=1), 0] = 1
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
wlp = os.path.join(d, "wl.txt")
open(wlp, "wb").write(b"\n".join(lut[w].tobytes() for w in wl) + b"\n")
L = 150
name = np.char.add("@r", np.char.zfill(np.arange(n).astype(str), 9)).astype("S").view(np.uint8).reshape(n, -1)
w = name.shape[1]
half = w + 1 + L + 3 + L + 1
rec = np.empty((n, 2 * half), dtype=np.uint8)
codes = rng.integers(0, 4, (n, 2 * L), dtype=np.uint8)
codes[:, :16] = wl[rng.integers(0, len(wl) // 20, n)]      # 5 K barcodes in use: ~200 pairs each
off = rng.random(n) < 0.1; p = rng.integers(0, 16, n); codes[off, p[off]] = (codes[off, p[off]] + 1) & 3
bases = lut[codes]
nn = rng.random(n) < 0.03; bases[nn, p[nn]] = ord("N")
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


def digest(path):
    return {f: hashlib.sha256(open(os.path.join(path, f), "rb").read()).hexdigest() for f in sorted(os.listdir(path))}


t = time.perf_counter()
ema_count.count_fastq(wlp, fq, os.path.join(d, "a"))
t1 = time.perf_counter()
st = ema_preproc.preproc_fastq(wlp, [os.path.join(d, "a.ema-ncnt")], os.path.join(d, "A"), fq, n_threads=8, n_buckets=nb)
t2 = time.perf_counter()
print(f"product: count {n / (t1 - t):,.0f} pairs/s, preproc {n / (t2 - t1):,.0f} pairs/s ({size / (t2 - t1) / 1e6:,.0f} MB/s in); {st}")
_out = os.environ.get("EMA_REF_OUT") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "ema_ref")
rc, rp = (os.path.join(_out, x) for x in ("ref_count", "ref_preproc"))
if os.path.exists(rc) and os.path.exists(rp):
    t = time.perf_counter()
    subprocess.run([rc, wlp, os.path.join(d, "b"), str(1 << 30), "0"], stdin=open(fq, "rb"), check=True, stderr=subprocess.DEVNULL)
    t1 = time.perf_counter()
    subprocess.run([rp, wlp, os.path.join(d, "B"), "0", str(10 << 20), "0", "8", str(nb), "0", os.path.join(d, "b.ema-ncnt")], stdin=open(fq, "rb"), check=True,
                   stderr=subprocess.DEVNULL)
    t2 = time.perf_counter()
    print(f"reference: count {n / (t1 - t):,.0f} pairs/s, preproc {n / (t2 - t1):,.0f} pairs/s ({size / (t2 - t1) / 1e6:,.0f} MB/s in); "
          f"bucket files identical: {digest(os.path.join(d, 'A')) == digest(os.path.join(d, 'B'))}")
