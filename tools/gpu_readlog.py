"""Dev aid (GPU): K2b's per-read log on one bench batch (EMA_PHASE_PROFILE=2) -> gpurun_out/<tag>_readlog.npy and a summary:
which reads the wave-per-read kernel spends its clocks on (by seed occurrences, chains, regions, DPs).
  EMA_PHASE_PROFILE=2 python tools/gpu_readlog.py [tag] [--genome-mbp 3100]"""
import ctypes as C, glob, os, sys, tempfile
os.environ.setdefault("EMA_PHASE_PROFILE", "2")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd.engine import Engine, default_opts
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
reads = sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0]
z = np.load(reads)
o = default_opts(); o.batch_pairs = (len(z["off"]) - 1) // 2
bases, off = z["bases"], z["off"]
if os.environ.get("EMA_READLOG_ONE_SLICE"):      # one lean slice of the batch on its own: K2b's queue-dry / last-wave-out times are those of one launch
    n1 = o.batch_pairs // 3
    off = off[:2 * n1 + 1]; bases = bases[:off[-1]]
    o.n_streams = 1
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
eng.stage(bases, off)
eng.run(); eng.sync()
eng.timing()                       # prints the phase split, resets the phase counters
eng._L.ema_engine_debug_readlog.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
p, n = C.POINTER(C.c_int32)(), C.c_size_t()
eng._check(eng._L.ema_engine_debug_readlog(eng._h, C.byref(p), C.byref(n)), "readlog")      # warm-up run discarded
C.CDLL(None).free(p)
eng.run(serial=True); eng.sync()
tm = eng.timing()
eng._check(eng._L.ema_engine_debug_readlog(eng._h, C.byref(p), C.byref(n)), "readlog")
log = np.ctypeslib.as_array(p, shape=(n.value, 8)).copy()
out = os.path.join(R, "gpurun_out", f"{tag}_readlog.npy")
os.makedirs(os.path.dirname(out), exist_ok=True)
np.save(out, log)
clk = log[:, 7].astype(np.int64) * 16
print(f"{len(log)} reads through K2b of {2 * o.batch_pairs}; isolated ms: seed {tm['seed_ms']:.1f} extend {tm['extend_ms']:.1f} rescue {tm['rescue_ms']:.1f} final {tm['final_ms']:.1f} full tier {tm['full_tier_ms']:.1f}; total wave clocks {clk.sum():.3e}")
print(f"handed over by K2a (chains ready): {(log[:, 1] < 0).sum()}  clocks share {clk[log[:, 1] < 0].sum() / clk.sum():.3f}")
for name, col, edges in (("seed occurrences", 2, [0, 8, 40, 200, 1000, 5000, 1 << 31]), ("chains", 3, [0, 4, 16, 64, 256, 1024, 1 << 31]),
                         ("regions before dedup", 5, [0, 2, 4, 8, 16, 64, 1 << 31]), ("extension DPs", 6, [0, 1, 3, 6, 12, 48, 1 << 31])):
    print(name)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (log[:, col] >= lo) & (log[:, col] < hi)
        print(f"   [{lo:5d}, {hi:10d})  reads {m.sum():8d} ({m.mean():6.3f})  clocks share {clk[m].sum() / clk.sum():6.3f}  mean clocks {clk[m].mean() if m.any() else 0:10.0f}")
srt = np.sort(clk)[::-1]
for f in (0.001, 0.01, 0.1):
    k = max(1, int(len(srt) * f))
    print(f"heaviest {f:.1%} of the reads: {srt[:k].sum() / clk.sum():.3f} of the clocks; the heaviest read {srt[0]:.0f} clocks")
eng.close()
