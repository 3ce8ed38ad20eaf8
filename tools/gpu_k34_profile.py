"""Dev aid (GPU): where K3's and K4's wavefronts spend their lifetimes on one bench batch -- the `make prof-lib` build of the library
(ema_amd/libema_engine_prof.so: phase clocks in scalar registers, ema_amd/csrc/dev_prof.hpp), slices in turn and overlapping.
  make prof-lib && python tools/gpu_k34_profile.py"""
import glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["EMA_ENGINE_LIB"] = "libema_engine_prof.so"
sys.path.insert(0, R)
import numpy as np
from ema_amd.engine import Engine, default_opts
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
o = default_opts(); o.batch_pairs = (len(z["off"]) - 1) // 2
eng = Engine(os.path.join(wd, "ref.fa"), opts=o)
eng.stage(z["bases"], z["off"])
eng.run(); eng.sync(); eng.timing()      # warm-up; prints and resets
print("--- one batch, slices one after another (isolated) ---", file=sys.stderr, flush=True)
eng.run(serial=True); eng.sync()
tm = eng.timing()
print(f"isolated ms per slice: seed {tm['seed_ms']:.1f} extend {tm['extend_ms']:.1f} rescue {tm['rescue_ms']:.1f} final {tm['final_ms']:.1f}; full tier {tm['full_tier_ms']:.1f}")
print("--- one batch, slices overlapping ---", file=sys.stderr, flush=True)
eng.run(); eng.sync()
tm = eng.timing()
print(f"overlapped ms per slice: seed {tm['seed_ms']:.1f} extend {tm['extend_ms']:.1f} rescue {tm['rescue_ms']:.1f} final {tm['final_ms']:.1f}; full tier {tm['full_tier_ms']:.1f}")
eng.close()
