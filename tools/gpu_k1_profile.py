"""Dev aid (GPU): K1 on one bench batch at the bench's scale.
  python tools/gpu_k1_profile.py            product build: isolated ms per slice (what rocprofv3 --kernel-trace is wrapped around)
  EMA_PHASE_PROFILE=1 python tools/gpu_k1_profile.py      diagnostic build: wave-ticks, busy lanes per tick, clocks per tick on stderr
  python tools/gpu_k1_profile.py dist [N]   CPU: the oracle's extends per read on N reads of the batch (how heavy the tail is)
Environment knobs (EMA_SEED_TAIL, EMA_KMER_K, EMA_SEED_PARK, EMA_SEED_ROUNDS, EMA_SEED_BLOCKS_PER_CU, LEAN_EXTENDS) apply."""
import glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
prefix = os.path.join(wd, "ref.fa")
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
if len(sys.argv) > 1 and sys.argv[1] == "dist":
    import oracle_lib as O
    from ema_amd import synth
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    pairs = synth.Pairs(z["bases"], z["off"])
    idx, opt = O.Index(prefix), O.default_opt()
    ext = np.zeros(n, dtype=np.int64)
    for r in range(n):
        O.stats_reset()
        O.collect_intv(idx, opt, pairs.read(r))
        ext[r] = O.stats_get()["n_ext"]
    print(f"{n} reads: extends per read mean {ext.mean():.1f} median {np.median(ext):.0f}")
    for q in (50, 75, 90, 95, 98, 99, 99.5, 99.9, 100):
        print(f"  p{q:<5} {np.percentile(ext, q):8.0f}")
    for b in (512, 1024, 1536, 2048, 3072, 4096, 8192):
        over = ext > b
        print(f"  budget {b:5d}: {100.0 * over.mean():6.3f} % of reads over it; they hold {100.0 * np.minimum(ext, b)[over].sum() / ext.clip(max=b).sum():5.2f} % of the budgeted extends; "
              f"extends under the budget {np.minimum(ext, b).sum() / n:7.1f} per read")
    sys.exit(0)
from ema_amd.engine import Engine, default_opts
o = default_opts(); o.batch_pairs = (len(z["off"]) - 1) // 2
if os.environ.get("LEAN_EXTENDS"):
    o.lean_seed_extends = int(os.environ["LEAN_EXTENDS"])
eng = Engine(prefix, opts=o)
eng.stage(z["bases"], z["off"])
eng.run(); eng.sync(); eng.timing()
for it in range(2):
    print("--- serial pass ---", file=sys.stderr, flush=True)
    eng.run(serial=True); eng.sync()
    tm = eng.timing()
    print(f"isolated ms per slice: seed {tm['seed_ms']:.2f} extend {tm['extend_ms']:.2f} rescue {tm['rescue_ms']:.2f} final {tm['final_ms']:.2f}; "
          f"full tier {tm['full_tier_ms']:.2f} {[round(x, 2) for x in tm['full_ms']]}", flush=True)
b = eng.fetch(allow_limit=True)
st = b.status
print(f"reads flagged long by the lean budget: {int(((st & 256) != 0).sum())}; pairs redone {b.n_redone}; any capacity flag {int((st & 127).any())}", flush=True)
eng.close()
