"""Dev aid (GPU): bucket files -> SAM text end to end at the default scale (BASELINE configs[2]'s shape, scaled down): N
bucket files of P pairs each, simulated from the bench workdir's genome with ~200 pairs per barcode, through ONE call of
ema_stream_sam (reader, engine, append stage, clouds / EM / duplicates, formatter) to /dev/null.  Prints pairs/s end to end and
the wall seconds inside every stage.
  python tools/gpu_sam_rate.py [N_BUCKETS] [PAIRS_PER_BUCKET]"""
import glob, json, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd import stream, synth
from ema_amd.engine import Engine, default_opts
n_b = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n_p = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
flat = np.load(os.path.join(wd, "genome.npy"), mmap_mode="r")
lens = json.loads(open(os.path.join(wd, "ref.fa.gstamp")).read())["lens"]
ctg, at = [], 0
for n in lens:
    ctg.append(flat[at:at + n]); at += n
t = time.time()
paths = []
for k in range(n_b):
    pairs = synth.make_pairs(ctg, n_p, seed=4242 + k, flat=flat)
    path = os.path.join(wd, f"ema-bin-{k:03d}")
    # one line per pair, as `ema preproc` writes them (vectorised: the per-pair Python writer is too slow at this size)
    l1, l2 = 127, 150
    reads = pairs.bases.reshape(n_p, l1 + l2)
    ids = np.char.add("@s", np.arange(n_p).astype(str)).astype("S")
    q1, q2 = b"F" * l1, b"F" * l2
    with open(path, "wb") as f:
        bc = pairs.barcodes
        for i in range(n_p):
            f.write(bc[i].tobytes() + b" " + ids[i] + b" " + reads[i, :l1].tobytes() + b" " + q1 + b" " + reads[i, l1:].tobytes() + b" " + q2 + b"\n")
    paths.append(path)
print(f"{n_b} buckets x {n_p} pairs written in {time.time() - t:.1f}s", flush=True)
o = default_opts(); o.batch_pairs = max(int(os.environ.get('EMA_SAM_BATCH', 1048576)), n_p)      # the stream lays small buckets end to end up to this
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
fd = os.open("/dev/null", os.O_WRONLY)
stream.stream_sam(eng, paths[:1], fd, rg_id=b"rg1")      # warm-up: buffers, page cache
import resource
stream.host_cpu_seconds(reset=True)
r0 = resource.getrusage(resource.RUSAGE_SELF)
t0 = time.perf_counter()
rep = int(os.environ.get("EMA_SAM_REPEAT", 1))      # the same files again and again: a longer stream without writing more of them
paths = paths * rep
bst, sst = stream.stream_sam(eng, paths, fd, rg_id=b"rg1", continue_cloud_ids=True)
dt = time.perf_counter() - t0
r1 = resource.getrusage(resource.RUSAGE_SELF)
print(f"host CPU seconds / wall second during the call: {(r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / dt:.1f}")
os.close(fd)
eng.close()
tot = n_b * n_p * rep; n_b *= rep
print(f"bucket files -> SAM text: {tot / dt:.0f} pairs/s end to end ({dt:.2f}s for {tot} pairs in {n_b} buckets)")
for name, key, src in (("reader", "read_s", bst), ("engine (stage+kernels+fetch)", "align_s", bst), ("append stage", "append_s", bst),
                       ("clouds/EM/duplicates", "select_s", sst), ("formatter + write", "write_s", sst)):
    v = [s[key] for s in src]
    print(f"  {name:30s} {sum(v):7.2f}s in all, {n_p / (sum(v) / n_b):12.0f} pairs/s inside the stage")
cpu = stream.host_cpu_seconds()
print("  host CPU seconds per million pairs, by stage: " + ", ".join(f"{k} {v / (tot / 1e6):.3f}" for k, v in cpu.items()) +
      f"; all stages {sum(cpu.values()) / (tot / 1e6):.3f}, the process {(r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / (tot / 1e6):.3f}")
print("  SAM statistics of bucket 0:", {k: v for k, v in sst[0].items() if k not in ("select_s", "write_s")})
