"""Host-to-host throughput of the boundary (dev aid; numbers quoted in DESIGN.md): ASCII reads in host memory in,
candidate lists + append_alignments records in host memory out, i.e. staging (nt4 + packing + H2D), all kernels,
D2H + assembly, and the host stage, for a stream of batches.  Two engines sharing one index take alternate batches
from two host threads, so that one batch's staging/fetching overlaps the other's kernels."""
import os, sys, time, argparse, tempfile, threading, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, __graft_entry__
__graft_entry__.ensure_built()
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1048576)
ap.add_argument("--batches", type=int, default=6, help="per engine")
ap.add_argument("--engines", type=int, default=2)
ap.add_argument("--genome-mbp", type=float, default=0.0)
a = ap.parse_args()
wd = os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid()); os.makedirs(wd, exist_ok=True)
prefix, pairs, _ = bench.build_workload(a, 0, 1, wd)
from ema_amd import engine as E
o = E.default_opts(); o.batch_pairs = a.pairs
engs = [E.Engine(prefix, opts=o)]
for _ in range(a.engines - 1):
    engs.append(E.Engine(None, opts=o, share=engs[0]))
L = engs[0]._L
L.ema_batch_append_alignments.argtypes = [C.POINTER(E.BatchOut), C.c_void_p, C.POINTER(E.Opts), C.c_double, C.POINTER(C.POINTER(E.AlnOut))]
L.ema_aln_free.argtypes = [C.POINTER(E.AlnOut)]
n_out = [0] * a.engines


def worker(k, n_batches):
    eng = engs[k]
    for _ in range(n_batches):
        eng.stage(pairs.bases, pairs.off)
        eng.run(); eng.sync()
        p = C.POINTER(E.BatchOut)()
        rc = L.ema_engine_fetch(eng._h, C.byref(p))
        assert rc == 0, rc
        q = C.POINTER(E.AlnOut)()
        assert L.ema_batch_append_alignments(p, pairs.off.ctypes.data, C.byref(o), 0.001, C.byref(q)) == 0
        n_out[k] += q.contents.n
        L.ema_aln_free(q); L.ema_batch_free(p)


for k in range(a.engines):      # warm-up (pinned buffers, first launches)
    worker(k, 1)
n_out = [0] * a.engines
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k, a.batches)) for k in range(a.engines)]
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
tot = a.pairs * a.batches * a.engines
print(f"end to end: {a.engines} engine(s) x {a.batches} batches of {a.pairs} pairs in {dt:.3f} s = {tot / dt / 1e6:.2f} M pairs/s "
      f"({sum(n_out)} records)", flush=True)
