"""Rate of the bucket reader (dev aid; numbers quoted in DESIGN.md): ema_bucket_read on the host's cores against the
oracle's one-thread, line-at-a-time restatement of the reference's read_special_fastq on the same shuffled bucket."""
import os, sys, time, argparse, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__
__graft_entry__.ensure_built()
from ema_amd import ingest
import oracle_lib as O
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1000000)
ap.add_argument("--no-oracle", action="store_true")
ap.add_argument("--device", action="store_true", help="ema_bucket_read_device (needs a GPU), with its phase times")
a = ap.parse_args()
rng = np.random.default_rng(3)
n, l1, l2 = a.pairs, 127, 150
n_bc = max(1, n // 200)
bcs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n_bc, 16))]
which = rng.integers(0, n_bc, n)
ids = np.char.add("@s", np.arange(n).astype(str)).astype("S")
w = max(len(x) for x in ids[-1:]) if n else 3
idw = ids.dtype.itemsize
line_len = 16 + 1 + idw + 1 + l1 + 1 + l1 + 1 + l2 + 1 + l2 + 1
buf = np.full((n, line_len), ord(" "), np.uint8)
buf[:, :16] = bcs[which]
idb = np.frombuffer(ids.tobytes(), np.uint8).reshape(n, idw)
c = 17
buf[:, c:c + idw] = np.where(idb == 0, ord("_"), idb); c += idw + 1      # identifiers padded to one width
buf[:, c:c + l1] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, l1))]; c += l1 + 1
buf[:, c:c + l1] = ord("F"); c += l1 + 1
buf[:, c:c + l2] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, l2))]; c += l2 + 1
buf[:, c:c + l2] = ord("F"); c += l2
buf[:, c] = ord("\n")
path = os.path.join(tempfile.gettempdir(), "ema_ingest_%d.fq" % os.getuid())
buf.tofile(path)
size = os.path.getsize(path)
del buf
ingest.read_bucket(path)      # page cache + first-touch
import ctypes as C
L = ingest._lib()
ts = []
for _ in range(5):      # the C call alone: copying the result into numpy arrays is the mirror's business
    p = C.POINTER(ingest._Bucket)()
    t = time.perf_counter(); rc = L.ema_bucket_read(path.encode(), 16, 0, 255, C.byref(p)); ts.append(time.perf_counter() - t)
    assert rc == 0
    L.ema_bucket_free(p)
t_prod = sorted(ts)[2]
b = ingest.read_bucket(path)
print(f"bucket of {n} pairs, {size / 1e6:.0f} MB, {len(b.group_off) - 1} barcode groups", flush=True)
print(f"ema_bucket_read (median of 5): {t_prod * 1e3:.0f} ms = {n / t_prod / 1e6:.2f} M pairs/s, {size / t_prod / 1e9:.2f} GB/s "
      f"on {min(32, os.cpu_count())} host threads", flush=True)
if a.device:
    from ema_amd import engine as _E
    L.ema_bucket_read_device.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(ingest._Bucket))]
    ts = []
    for k in range(6):
        if k == 5:
            _E.set_tuning(ingest_prof=1)
        p = C.POINTER(ingest._Bucket)()
        t = time.perf_counter(); rc = L.ema_bucket_read_device(path.encode(), 16, 0, 255, 0, C.byref(p)); ts.append(time.perf_counter() - t)
        assert rc == 0 and p.contents.dev
        L.ema_bucket_free(p)
    _E.set_tuning()
    t_dev = sorted(ts[1:])[2]
    print(f"ema_bucket_read_device (median of 5 after the first): {t_dev * 1e3:.0f} ms = {n / t_dev / 1e6:.2f} M pairs/s, {size / t_dev / 1e9:.2f} GB/s", flush=True)
if not a.no_oracle:
    L = O.lib()
    import ctypes as C
    L.orc_read_special_fastq.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.POINTER(O.FastqRec)), C.POINTER(C.POINTER(O.FastqRec)), C.POINTER(C.c_size_t)]
    r1, r2, k = C.POINTER(O.FastqRec)(), C.POINTER(O.FastqRec)(), C.c_size_t()
    t = time.perf_counter()
    assert L.orc_read_special_fastq(path.encode(), 16, 0, C.byref(r1), C.byref(r2), C.byref(k)) == 0
    t_orc = time.perf_counter() - t
    # spot-check equality on a sample (the tests do it exhaustively on small buckets)
    for i in list(range(0, n, max(1, n // 1000))) + [n - 1]:
        assert r1[i].bc == int(b.bc[i]) and r1[i].read == b.read(2 * i) and r2[i].read == b.read(2 * i + 1) and r1[i].id == b.ident(i), i
    print(f"oracle (the reference's way: one thread, a line at a time): {t_orc * 1e3:.0f} ms = {n / t_orc / 1e6:.2f} M pairs/s, "
          f"{size / t_orc / 1e9:.2f} GB/s; ratio {t_orc / t_prod:.1f}x", flush=True)
os.remove(path)
from ema_amd import stream
print("reader CPU seconds per million pairs (all runs above): %.3f" % (stream.host_cpu_seconds()["reader"] / (5 * n / 1e6)))
