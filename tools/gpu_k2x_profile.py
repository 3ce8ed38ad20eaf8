"""Dev aid (GPU): K2x (k_ext_lane.hip, one lane per seed extension) on one bench batch -- its counters (tuning knob ext_lane_prof=1:
wavefront lifetimes, row-steps, lanes busy per row-step, DP cells, tasks, DP sides per class of task) and the isolated kernel
times per slice with the lane route on and off.
  python tools/gpu_k2x_profile.py [phase]      (phase: also K2b's product-build phase profile, as tools/gpu_k2_profile.py)"""
import ctypes as C, glob, os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
from ema_amd import engine as E
wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
phase = len(sys.argv) > 1 and sys.argv[1] == "phase"
for lane_route in (1, 0):
    E.set_tuning()
    E.set_tuning(ext_lane=lane_route, ext_lane_prof=lane_route, **({"phase_profile": 3} if phase else {}))
    o = E.default_opts(); o.batch_pairs = (len(z["off"]) - 1) // 2
    eng = E.Engine(os.path.join(wd, "ref.fa"), opts=o)
    eng.stage(z["bases"], z["off"])
    eng.run(); eng.sync(); eng.timing()
    L = E.load_library()
    L.ema_engine_debug_xprof.argtypes = [C.c_void_p, C.c_void_p]
    buf = (C.c_uint64 * 24)()
    L.ema_engine_debug_xprof(eng._h, buf)      # drop the warm-up's
    print(f"--- ext_lane={lane_route}: one batch, slices one after another (isolated) ---", file=sys.stderr, flush=True)
    eng.run(serial=True); eng.sync()
    tm = eng.timing()
    print(f"ext_lane={lane_route} isolated ms per slice: seed {tm['seed_ms']:.1f} extend {tm['extend_ms']:.1f} rescue {tm['rescue_ms']:.1f} final {tm['final_ms']:.1f}; full tier {tm['full_tier_ms']:.1f}")
    if lane_route:
        L.ema_engine_debug_xprof(eng._h, buf)
        v = list(buf)
        for k, name in enumerate(("query < 64", "query < 128", "query < 256")):
            life, waves, rowsteps, lanerows, cells, tasks, sides = v[8 * k:8 * k + 7]
            if waves:
                print(f"K2x {name:11s}: {tasks} tasks ({sides} DP sides), {cells} cells, {rowsteps} row-steps, {lanerows / max(1, rowsteps):.1f} lanes busy per row-step, "
                      f"{cells / max(1, lanerows):.1f} cells per lane-row, {cells / max(1, rowsteps) / 64:.2f} of the cell slots used (by mean band), "
                      f"lifetimes {life / 1e9:.2f} Gclk over {waves} wavefronts, {life / max(1, cells) * 64:.0f} clocks per 64 cells")
    print(f"--- ext_lane={lane_route}: one batch, slices overlapping ---", file=sys.stderr, flush=True)
    eng.run(); eng.sync()
    tm = eng.timing()
    print(f"ext_lane={lane_route} overlapped ms per slice: seed {tm['seed_ms']:.1f} extend {tm['extend_ms']:.1f} rescue {tm['rescue_ms']:.1f} final {tm['final_ms']:.1f}; full tier {tm['full_tier_ms']:.1f}")
    eng.close()
