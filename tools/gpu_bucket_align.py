"""Host-to-host rate of ema_engine_align_pairs on an input larger than the engine's batch capacity (dev aid; numbers
quoted in DESIGN.md): the library works in pieces, alternating over two sets of batch buffers from two host threads
(EMA_ALIGN_PIPELINE=0: one set, in sequence)."""
import os, sys, time, argparse, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, __graft_entry__
__graft_entry__.ensure_built()
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=1048576, help="engine batch capacity = pairs in the generated workload")
ap.add_argument("--copies", type=int, default=8, help="the bucket is the workload repeated this many times")
ap.add_argument("--genome-mbp", type=float, default=0.0)
a = ap.parse_args()
wd = os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid()); os.makedirs(wd, exist_ok=True)
prefix, pairs, _ = bench.build_workload(a, 0, 1, wd)
from ema_amd import engine as E
o = E.default_opts(); o.batch_pairs = a.pairs
eng = E.Engine(prefix, opts=o)
bases = np.tile(pairs.bases, a.copies)
step = int(pairs.off[-1])
off = np.concatenate([pairs.off[:-1] + k * step for k in range(a.copies)] + [np.array([a.copies * step], dtype=pairs.off.dtype)])
first = eng.align_pairs_any(pairs.bases, pairs.off)      # warm-up, one piece
import ctypes as C
bases = np.ascontiguousarray(bases, dtype=np.uint8); off = np.ascontiguousarray(off, dtype=np.uint32)
L = eng._L
L.ema_engine_align_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(E.BatchOut))]


def call():      # the C call alone is timed; copying the result into numpy arrays is the test's business
    p = C.POINTER(E.BatchOut)()
    t = time.perf_counter()
    rc = L.ema_engine_align_pairs(eng._h, bases.ctypes.data, off.ctypes.data, (len(off) - 1) // 2, C.byref(p))
    t = time.perf_counter() - t
    assert rc == 0, rc
    return eng._take(p), t


batch, dt1 = call()
batch2, dt2 = call()
t0, t1, t2 = 0.0, dt1, dt1 + dt2
n = a.pairs * a.copies
assert len(batch.cand_off) == 2 * n + 1
m, mc = len(first.cand), len(first.cigar)
assert len(batch.cand) == a.copies * m and len(batch.cigar) == a.copies * mc
for k in (0, a.copies - 1):      # every piece is the single batch again; only the offsets into the CIGAR pool move
    part = batch.cand[k * m:(k + 1) * m]
    for f in first.cand.dtype.names:
        want = first.cand[f] + (k * mc if f == "cigar_off" else 0)
        assert (part[f] == want).all(), (k, f)
    assert (batch.cigar[k * mc:(k + 1) * mc] == first.cigar).all(), k
assert (batch2.cand == batch.cand).all() and (batch2.cigar == batch.cigar).all()
print(f"bucket of {n} pairs, capacity {a.pairs}: first call {n / (t1 - t0) / 1e6:.2f} M pairs/s (creates the second buffer set), "
      f"second call {n / (t2 - t1) / 1e6:.2f} M pairs/s; pipeline={os.environ.get('EMA_ALIGN_PIPELINE', '1')}", flush=True)
eng.close()
