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
