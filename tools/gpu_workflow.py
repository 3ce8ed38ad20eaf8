"""Dev aid (GPU): the whole workflow through C-ABI calls only -- raw interleaved FASTQ -> ema_count_fastq -> ema_preproc_fastq
(bucket files) -> ema_stream_sam (reader, engine, append stage, clouds / EM / duplicates, formatter) -> SAM text on /dev/null -- at the
default scale, with the wall time of every step.  Uses the bench workdir's genome and index.
  python tools/gpu_workflow.py [N_PAIRS] [N_BUCKETS]"""
import json, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd import count as ema_count, preproc as ema_preproc, stream, synth
from ema_amd.engine import Engine, default_opts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 20
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
flat = np.load(os.path.join(wd, "genome.npy"), mmap_mode="r")
lens = json.loads(open(os.path.join(wd, "ref.fa.gstamp")).read())["lens"]
ctg, at = [], 0
for x in lens:
    ctg.append(flat[at:at + x]); at += x
t = time.time()
pairs = synth.make_pairs(ctg, n, seed=777, flat=flat)
l1, l2 = 127, 150
reads = np.asarray(pairs.bases).reshape(n, l1 + l2)
rng = np.random.default_rng(5)
# raw mate 1 = barcode (16) + 7 bases + read; 5 % of the barcodes get one base changed (preproc corrects them back)
bc = pairs.barcodes.copy()
hit = rng.random(n) < 0.05
pos = rng.integers(0, 16, n)
lut = np.frombuffer(b"ACGT", dtype=np.uint8)
alt = lut[rng.integers(0, 4, n)]
bc[hit, pos[hit]] = alt[hit]
name = np.char.add("@s", np.char.zfill(np.arange(n).astype(str), 8)).astype("S").view(np.uint8).reshape(n, -1)
w = name.shape[1]
L1 = 16 + 7 + l1
rec = np.empty((n, (w + 1 + L1 + 3 + L1 + 1) + (w + 1 + l2 + 3 + l2 + 1)), dtype=np.uint8)
c = 0
rec[:, c:c + w] = name; c += w
rec[:, c] = 10; c += 1
rec[:, c:c + 16] = bc; rec[:, c + 16:c + 23] = np.frombuffer(b"ACGTACG", dtype=np.uint8); rec[:, c + 23:c + L1] = reads[:, :l1]; c += L1
rec[:, c] = 10; rec[:, c + 1] = ord("+"); rec[:, c + 2] = 10; c += 3
rec[:, c:c + L1] = ord("F"); c += L1
rec[:, c] = 10; c += 1
rec[:, c:c + w] = name; c += w
rec[:, c] = 10; c += 1
rec[:, c:c + l2] = reads[:, l1:]; c += l2
rec[:, c] = 10; rec[:, c + 1] = ord("+"); rec[:, c + 2] = 10; c += 3
rec[:, c:c + l2] = ord("F"); c += l2
rec[:, c] = 10
d = tempfile.mkdtemp(prefix="ema_workflow_", dir=wd)
fq = os.path.join(d, "raw.fastq")
rec.tofile(fq)
wl = np.unique(pairs.barcodes.view("S16").ravel())
wlp = os.path.join(d, "wl.txt")
open(wlp, "wb").write(b"\n".join(wl.tolist()) + b"\n")
size = os.path.getsize(fq)
print(f"{n} pairs simulated and written as raw interleaved FASTQ ({size / 1e9:.2f} GB, {len(wl)} barcodes) in {time.time() - t:.1f}s", flush=True)
o = default_opts()
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
t0 = time.perf_counter()
cst = ema_count.count_fastq(wlp, fq, os.path.join(d, "c"))
t1 = time.perf_counter()
pst = ema_preproc.preproc_fastq(wlp, [os.path.join(d, "c.ema-ncnt")], os.path.join(d, "b"), fq, n_threads=16, n_buckets=nb)
t2 = time.perf_counter()
paths = [os.path.join(d, "b", f"ema-bin-{k:03d}") for k in range(nb)]
fd = os.open("/dev/null", os.O_WRONLY)
bst, sst = stream.stream_sam(eng, paths, fd, rg_id=b"rg1", continue_cloud_ids=True)
t3 = time.perf_counter()
os.close(fd); eng.close()
lines = sum(s["lines"] for s in sst)
print(f"count   {n / (t1 - t0):12,.0f} pairs/s ({t1 - t0:.2f}s)  {cst}")
print(f"preproc {n / (t2 - t1):12,.0f} pairs/s ({t2 - t1:.2f}s)  {pst}")
print(f"buckets -> SAM {pst['pairs_written'] / (t3 - t2):12,.0f} pairs/s ({t3 - t2:.2f}s)  {lines} SAM lines")
print(f"raw FASTQ -> SAM text, the three calls one after another: {n / (t3 - t0):,.0f} pairs/s ({t3 - t0:.2f}s for {n} pairs)")
assert pst["pairs_written"] + pst["pairs_nobc"] + pst["pairs_skipped"] == n and lines == 2 * pst["pairs_written"]
for f in os.listdir(os.path.join(d, "b")):
    os.remove(os.path.join(d, "b", f))
os.remove(fq)
