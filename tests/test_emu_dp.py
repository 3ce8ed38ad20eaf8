"""The global DP's band layout and the traceback's runs (ema_amd/csrc/dev_dp.hpp) through the host SIMT interpreter (tests/emu: the
kernel sources compiled for the CPU, 64 lanes in lockstep) against the oracle's ksw_global2 -- the CPU-side check of the two
layouts; the parity evidence proper is tests/test_gpu_dp.py on the GPU."""
import ctypes as C

import numpy as np

import dp_cases as D
import emu_lib


def test_global_band_layout_and_traceback_runs_under_the_interpreter():
    L = emu_lib.lib()
    rng = np.random.default_rng(106)
    qs, ts, prm = D.band_cases(rng, 250)
    band = sum(1 for q, t, w in zip(qs, ts, prm) if 2 * w + 1 <= 64 and abs(len(t) - len(q)) <= w)
    assert band > 200      # the cases do go through the band layout
    qb, qo = D.flat(qs); tb, to = D.flat(ts)
    n, cap = len(qs), 640
    out = np.zeros((n, 2), np.int32); cig = np.zeros((n, cap), np.uint32)
    L.emu_dp_global.restype = None
    L.emu_dp_global.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.emu_dp_global(qb.ctypes.data, qo.ctypes.data, tb.ctypes.data, to.ctypes.data, prm.ctypes.data, n, out.ctypes.data, cig.ctypes.data, cap)
    bad = []
    for i in range(n):
        sc, ops = D.oracle_global(qs[i], ts[i], prm[i])
        if sc != out[i, 0] or ops != cig[i, :max(0, out[i, 1])].tolist():
            bad.append(i)
    assert not bad, f"{len(bad)} global tasks differ, first {[(len(qs[i]), len(ts[i]), int(prm[i])) for i in bad[:5]]}"


def test_wavefront_introsort_equals_the_single_lane_one_under_the_interpreter():
    """dev_sort.hpp, ema_introsort_wave: klib's ks_introsort run by a whole wavefront (a Hoare scan as a pairing of two stopper
    sequences read off ballots, the closing insertion sort as a stable ranking) must leave EXACTLY the array the single-lane form
    leaves -- ties included, that is what bwa's chain filter sees.  Sizes 1..256; keys with a handful of distinct weights (ties
    everywhere: chain weights are seed lengths), with distinct ones, sorted, reversed, all equal, organ-pipe; the filter's
    comparison (high words, descending) and a plain ascending one."""
    L = emu_lib.lib()
    L.emu_sort.restype = None
    L.emu_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(211)
    cap = 256
    sizes = list(range(1, 41)) + [63, 64, 65, 66, 100, 127, 128, 129, 130, 191, 192, 193, 200, 255, 256] + rng.integers(17, 257, 60).tolist()
    tasks = []
    for n in sizes:
        kinds = [rng.integers(19, 19 + max(1, n // 6), n), rng.integers(19, 151, n), rng.permutation(n) + 1, np.arange(n) + 1, np.arange(n)[::-1] + 1,
                 np.full(n, 42), np.concatenate([np.arange((n + 1) // 2), np.arange(n // 2)[::-1]]) + 1, rng.integers(19, 23, n)]
        for w in kinds:
            tasks.append((np.asarray(w, np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64))
    for by_weight in (1, 0):
        a = np.zeros((len(tasks), cap), np.uint64)
        for t, k in enumerate(tasks):
            a[t, :len(k)] = k if by_weight else (k >> np.uint64(32)) * np.uint64(1 if t % 2 else 1 << 20) + (k & np.uint64(0xffffffff)) * np.uint64(t % 2)
        b = a.copy()
        n = np.array([len(k) for k in tasks], np.int32)
        L.emu_sort(a.ctypes.data, b.ctypes.data, n.ctypes.data, cap, len(tasks), by_weight)
        bad = [t for t in range(len(tasks)) if not (a[t] == b[t]).all()]
        assert not bad, f"{len(bad)} of {len(tasks)} arrays differ (by_weight={by_weight}), first: n = {int(n[bad[0]])}, task {bad[0]}"
        for t in range(0, len(tasks), 7):      # ... and both are sorted
            k = (b[t, :n[t]] >> np.uint64(32)) if by_weight else b[t, :n[t]]
            assert (np.diff(k.astype(np.int64)) * (-1 if by_weight else 1) >= 0).all()
