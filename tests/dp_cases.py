"""Random DP task generators shared by the emulator checks and the GPU parity tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

import oracle_lib as O


def mutate(rng, s, sub=0.05, indel=0.02, n_rate=0.0):
    out = []
    for b in s:
        r = rng.random()
        if r < indel / 2:
            continue
        if r < indel:
            out.append(int(rng.integers(0, 4)))
        if rng.random() < sub:
            b = (b + int(rng.integers(1, 4))) & 3
        if rng.random() < n_rate:
            b = 4
        out.append(int(b))
    return np.array(out, dtype=np.uint8)


def flat(seqs):
    off = np.zeros(len(seqs) + 1, dtype=np.uint32)
    off[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.concatenate(seqs).astype(np.uint8) if len(seqs) else np.zeros(0, np.uint8)
    return np.ascontiguousarray(buf), off


def extend_cases(rng, n, max_q=250):
    qs, ts, prm = [], [], []
    for _ in range(n):
        ql = int(rng.integers(1, max_q + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        kind = rng.random()
        if kind < 0.15:    # target starts with the query itself (the kernel's exact shortcut), with or without a tail
            t = np.concatenate([q, rng.integers(0, 4, int(rng.choice([0, 0, 1, 7, 60]))).astype(np.uint8)])
        elif kind < 0.35:  # ... or differs from it in exactly one position (the one-mismatch shortcut): anywhere, often near an end
            t = q.copy()
            pos = int(rng.choice([0, ql - 1, max(0, ql - 5), max(0, ql - 6), max(0, ql - 7), int(rng.integers(0, ql))]))
            t[pos] = (t[pos] + int(rng.integers(1, 4))) & 3
            t = np.concatenate([t, rng.integers(0, 4, int(rng.choice([0, 0, 1, 7, 60]))).astype(np.uint8)])
        elif kind < 0.6:
            t = mutate(rng, q, sub=rng.choice([0.0, 0.02, 0.1]), indel=rng.choice([0.0, 0.01, 0.05]))
            t = np.concatenate([t, rng.integers(0, 4, int(rng.integers(0, 120))).astype(np.uint8)])
        elif kind < 0.8:   # big indel
            cut = int(rng.integers(0, ql))
            gap = rng.integers(0, 4, int(rng.integers(1, 60))).astype(np.uint8)
            t = np.concatenate([q[:cut], gap, q[cut:], rng.integers(0, 4, 30).astype(np.uint8)])
        else:
            t = rng.integers(0, 4, int(rng.integers(1, 300))).astype(np.uint8)
        if rng.random() < 0.2:
            q = q.copy(); q[rng.random(ql) < 0.03] = 4
        if len(t) == 0:
            t = np.array([0], dtype=np.uint8)
        qs.append(q); ts.append(t)
        prm.append([int(rng.choice([100, 200, 5, 30])), 5, int(rng.choice([100, 0, 20, 3])), int(rng.integers(1, 150))])
    return qs, ts, np.array(prm, dtype=np.int32)


def oracle_extend(q, t, prm, opt=None):
    L = O.lib(); opt = opt or O.default_opt()
    vals = [C.c_int() for _ in range(5)]
    sc = L.orc_ksw_extend2(len(q), q.tobytes(), len(t), t.tobytes(), 5, opt.mat, opt.o_del, opt.e_del, opt.o_ins,
                           opt.e_ins, int(prm[0]), int(prm[1]), int(prm[2]), int(prm[3]), *[C.byref(v) for v in vals])
    return [sc] + [v.value for v in vals]


def global_cases(rng, n, max_q=250):
    qs, ts, prm = [], [], []
    for _ in range(n):
        ql = int(rng.integers(1, max_q + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        t = mutate(rng, q, sub=rng.choice([0.0, 0.03, 0.1]), indel=rng.choice([0.0, 0.01, 0.04]))
        if rng.random() < 0.2:
            cut = int(rng.integers(0, len(t) + 1))
            t = np.concatenate([t[:cut], rng.integers(0, 4, int(rng.integers(1, 40))).astype(np.uint8), t[cut:]])
        if len(t) == 0:
            t = np.array([1], dtype=np.uint8)
        if rng.random() < 0.2:
            q = q.copy(); q[rng.random(ql) < 0.03] = 4
        d = abs(len(t) - ql)
        w = d + 3 + int(rng.integers(0, 60))
        qs.append(q); ts.append(t); prm.append(w)
    return qs, ts, np.array(prm, dtype=np.int32)


def oracle_global(q, t, w, opt=None):
    L = O.lib(); opt = opt or O.default_opt()
    n = C.c_int(); cig = C.POINTER(C.c_uint32)()
    sc = L.orc_ksw_global2(len(q), q.tobytes(), len(t), t.tobytes(), 5, opt.mat, opt.o_del, opt.e_del, opt.o_ins,
                           opt.e_ins, int(w), C.byref(n), C.byref(cig))
    ops = [cig[i] for i in range(n.value)]
    C.CDLL(None).free(cig)
    return sc, ops


def local_cases(rng, n, max_q=250):
    qs, ts, prm = [], [], []
    for _ in range(n):
        ql = int(rng.integers(20, max_q + 1))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        tl = int(rng.integers(30, 800))
        t = rng.integers(0, 4, tl).astype(np.uint8)
        kind = rng.random()
        if kind < 0.7:      # plant (part of) the query, possibly twice
            for _rep in range(int(rng.integers(1, 3))):
                a = int(rng.integers(0, ql // 2)); b = int(rng.integers(a + 10, ql + 1))
                piece = mutate(rng, q[a:b], sub=rng.choice([0.0, 0.03, 0.1]), indel=rng.choice([0.0, 0.02]))
                p = int(rng.integers(0, max(1, tl - len(piece))))
                t[p:p + len(piece)] = piece[:tl - p]
        if rng.random() < 0.2:
            q = q.copy(); q[rng.random(ql) < 0.02] = 4
        byte = ql < 250
        qs.append(q); ts.append(t)
        prm.append([16 if byte else 8, 19, 0x10000])
    return qs, ts, np.array(prm, dtype=np.int32)


def oracle_local_pass(q, t, p, minsc, endsc, opt=None):
    """One pass of the local kernel = orc_ksw_align2 without XSTART (XSUBO|minsc, optional XSTOP)."""
    L = O.lib(); opt = opt or O.default_opt()
    xtra = (0x10000 if p == 16 else 0)
    if minsc < 0x10000:
        xtra |= 0x40000 | minsc
    if endsc < 0x10000:
        xtra = (0x10000 if p == 16 else 0) | 0x20000 | endsc
    r = L.orc_ksw_align2(len(q), q.tobytes(), len(t), t.tobytes(), 5, opt.mat, opt.o_del, opt.e_del, opt.o_ins,
                         opt.e_ins, xtra)
    return [r.score, r.te, r.qe, r.score2, r.te2]


def scoring(a, b, o_del, e_del, o_ins, e_ins):
    """(engine options, oracle options) with bwa's -A -B -O -E set (bwa_fill_scmat for the oracle's matrix)."""
    from ema_amd.engine import default_opts
    eo, oo = default_opts(), O.default_opt()
    for o in (eo, oo):
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = a, b, o_del, e_del, o_ins, e_ins
    for i in range(5):
        for j in range(5):
            oo.mat[i * 5 + j] = -1 if i == 4 or j == 4 else (a if i == j else -b)
    return eo, oo


def band_cases(rng, n):
    """Global-alignment tasks for the band layout of the global DP (dev_dp.hpp, ema_wave_global_band: 2w + 1 <= 64 and
    |tlen - qlen| <= w) and the traceback's runs: bands from 0 (a single diagonal) to 31, queries from 1 base to 255 at the sizes
    where the column layout changes shape, clean diagonals, gaps of 1-24 bases, ambiguous bases in either sequence."""
    qs, ts, prm = [], [], []
    for _ in range(n):
        ql = int(rng.choice([1, 2, 3, 5, 17, 40, 63, 64, 65, 100, 127, 150, 200, 250, 255, int(rng.integers(1, 256))]))
        q = rng.integers(0, 4, ql).astype(np.uint8)
        t = mutate(rng, q, sub=rng.choice([0.0, 0.03, 0.1, 0.3]), indel=rng.choice([0.0, 0.01, 0.05]))
        if rng.random() < 0.3:
            cut = int(rng.integers(0, len(t) + 1))
            t = np.concatenate([t[:cut], rng.integers(0, 4, int(rng.integers(1, 25))).astype(np.uint8), t[cut:]])
        if rng.random() < 0.2 and len(t) > 8:
            cut = int(rng.integers(0, len(t) - 6))
            t = np.concatenate([t[:cut], t[cut + int(rng.integers(1, 6)):]])
        if len(t) == 0:
            t = np.array([2], np.uint8)
        if rng.random() < 0.2:
            q = q.copy(); q[rng.random(ql) < 0.05] = 4
        if rng.random() < 0.1:
            t = t.copy(); t[rng.random(len(t)) < 0.05] = 4
        d = abs(len(t) - ql)
        w = d + int(rng.choice([0, 0, 1, 3, 3, 5, 10, 20]))
        if rng.random() < 0.9:
            w = max(w, 1)
        qs.append(q); ts.append(t); prm.append(w)
    return qs, ts, np.array(prm, dtype=np.int32)
