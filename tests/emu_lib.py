"""ctypes binding of tests/emu/_build/libemu.so (host SIMT interpreter running the HIP kernel sources).
Debugging aid for machines without a GPU; the parity evidence is the `-m gpu` suite."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib(rebuild=False):
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "tests", "emu", "_build", "libemu.so")
        srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("harness.cpp", "simt_emu.h")]
        csrc = os.path.join(ROOT, "ema_amd", "csrc")
        srcs += [os.path.join(csrc, f) for f in os.listdir(csrc)]
        if rebuild or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build.sh")])
        L = C.CDLL(os.environ.get("EMU_SO", so))      # EMU_SO: an alternative build (e.g. with a small EMA_OCC_SUPER_SHIFT)
        L.emu_index_load.restype = C.c_void_p
        L.emu_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.emu_index_free.argtypes = [C.c_void_p]
        L.emu_seed.restype = C.c_int
        L.emu_seed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.emu_align.restype = C.c_int
        L.emu_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def index_load(prefix):
    err = C.create_string_buffer(512)
    h = lib().emu_index_load(prefix.encode(), err, 512)
    if not h:
        raise RuntimeError(err.value.decode())
    return h


def seed(h, nt4_bases: np.ndarray, off: np.ndarray, n_blocks=1, cap=512, wave=False):
    """K1 through the interpreter (wave=True: K1w, one wavefront per read)."""
    n_reads = len(off) - 1
    intv = np.zeros((n_reads, cap, 4), dtype=np.uint64)
    n_intv = np.zeros(n_reads, dtype=np.int32)
    status = np.zeros(n_reads, dtype=np.int32)
    if wave:
        lib().emu_seed_wave.restype = C.c_int
        lib().emu_seed_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        got = lib().emu_seed_wave(h, nt4_bases.ctypes.data, off.ctypes.data, n_reads, intv.ctypes.data, n_intv.ctypes.data,
                                  status.ctypes.data)
    else:
        got = lib().emu_seed(h, nt4_bases.ctypes.data, off.ctypes.data, n_reads, intv.ctypes.data, n_intv.ctypes.data,
                             status.ctypes.data, n_blocks)
    assert got == cap
    return intv, n_intv, status


REG_DTYPE = np.dtype([("rb", "<i8"), ("re", "<i8")] +
                     [(n, "<i4") for n in ("qb", "qe", "rid", "score", "truesc", "sub", "csub", "w", "seedcov", "secondary",
                                           "seedlen0", "n_comp", "is_alt")] + [("frac_rep", "<f4")], align=True)


def align(h, nt4_bases, off, n_blocks=1, cap=256):
    """K1 + K2 through the interpreter: per-read regions (structured array [n_reads, cap]), counts, status."""
    n_reads = len(off) - 1
    assert lib().emu_sizeof_reg() == REG_DTYPE.itemsize
    regs = np.zeros((n_reads, cap), dtype=REG_DTYPE)
    n_regs = np.zeros(n_reads, dtype=np.int32)
    status = np.zeros(n_reads, dtype=np.int32)
    got = lib().emu_align(h, nt4_bases.ctypes.data, off.ctypes.data, n_reads, regs.ctypes.data, n_regs.ctypes.data,
                          status.ctypes.data, n_blocks)
    assert got == cap
    return regs, n_regs, status


ALN_DTYPE = np.dtype([("pos", "<i8"), ("is_rev", "<i4"), ("NM", "<i4"), ("n_cigar", "<i4"), ("cigar_off", "<u4")])


def pipeline(h, nt4_bases, off, upto=4, cap=256, cig_cap=1024):
    """K1..K4 through the interpreter.  Returns regs, n_regs, alns, cigars, cig_n, status (per read)."""
    n_reads = len(off) - 1
    L = lib()
    L.emu_pipeline.restype = C.c_int
    L.emu_pipeline.argtypes = [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int]
    regs = np.zeros((n_reads, cap), dtype=REG_DTYPE)
    n_regs = np.zeros(n_reads, dtype=np.int32)
    alns = np.zeros((n_reads, cap), dtype=ALN_DTYPE)
    cigars = np.zeros((n_reads, cig_cap), dtype=np.uint32)
    cig_n = np.zeros(n_reads, dtype=np.int32)
    status = np.zeros(n_reads, dtype=np.int32)
    got = L.emu_pipeline(h, nt4_bases.ctypes.data, off.ctypes.data, n_reads, regs.ctypes.data, n_regs.ctypes.data,
                         alns.ctypes.data, cigars.ctypes.data, cig_n.ctypes.data, status.ctypes.data, upto)
    assert got == cig_cap
    return regs, n_regs, alns, cigars, cig_n, status


def pipeline_tiers(h, nt4_bases, off, lean=(8, 2, 8), full_pairs=64, cap=256, cig_cap=1024):
    """The engine's two capacity tiers through the interpreter (lean = per-read interval/region/CIGAR capacities of the
    first tier).  Returns regs, n_regs, alns, cigars, cig_n, status, tier (0 lean, 1 full, -1 flagged and not redone), n_listed."""
    n_reads = len(off) - 1
    L = lib()
    L.emu_pipeline_tiers.restype = C.c_int
    L.emu_pipeline_tiers.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p] * 7
    regs = np.zeros((n_reads, cap), dtype=REG_DTYPE)
    n_regs = np.zeros(n_reads, dtype=np.int32)
    alns = np.zeros((n_reads, cap), dtype=ALN_DTYPE)
    cigars = np.zeros((n_reads, cig_cap), dtype=np.uint32)
    cig_n = np.zeros(n_reads, dtype=np.int32)
    status = np.zeros(n_reads, dtype=np.int32)
    tier = np.zeros(n_reads, dtype=np.int32)
    n = L.emu_pipeline_tiers(h, nt4_bases.ctypes.data, off.ctypes.data, n_reads, lean[0], lean[1], lean[2], full_pairs,
                             regs.ctypes.data, n_regs.ctypes.data, alns.ctypes.data, cigars.ctypes.data, cig_n.ctypes.data,
                             status.ctypes.data, tier.ctypes.data)
    return regs, n_regs, alns, cigars, cig_n, status, tier, n


class _SamJob(C.Structure):      # csrc/dev_sam.h
    _fields_ = [(n, C.c_void_p) for n in ("bases", "quals", "off", "ids", "id_off", "bc", "cigar", "desc", "xa", "sel_at", "names", "name_off", "rg", "bx")] + \
               [("n_lines", C.c_uint32), ("cigar_lo", C.c_uint32)] + \
               [(n, C.c_int32) for n in ("has_rg", "rg_len", "bx_len", "bc_len", "is_haplotag", "insert_min", "insert_max")]


def sam_format(bk, cigar_ptr, cigar_lo, descs, xas, sel_at, n_sel, contig_names, opts) -> bytes:
    """k_sam.hip's three kernels under the interpreter.  bk: a ctypes ema_bucket; cigar_ptr: address of operation cigar_lo of the
    batch's CIGAR array; descs / xas / sel_at: addresses (ema_sam_desc[], ema_sam_xa[], uint32[]); opts: ema_amd.sam.SamOpts."""
    L = lib()
    L.emu_sam_format.restype = C.c_longlong
    L.emu_sam_format.argtypes = [C.POINTER(_SamJob), C.c_void_p, C.c_longlong]
    assert L.emu_sizeof_sam_job() == C.sizeof(_SamJob)
    names = b"".join(contig_names) + b"\0" * 8
    name_off = np.cumsum([0] + [len(n) for n in contig_names]).astype(np.uint32)
    rg = opts.rg_id
    rg_len = 0
    if rg is not None:
        for i, c in enumerate(rg):
            if c == 0x20 or 9 <= c <= 13:
                break
            rg_len = i + 1
    rg_buf, bx_buf, names_buf = C.create_string_buffer(rg or b"", max(8, len(rg or b"") + 8)), C.create_string_buffer(opts.bx_index, len(opts.bx_index) + 8), C.create_string_buffer(names, len(names))
    J = _SamJob()
    adr = lambda p: C.cast(p, C.c_void_p).value if p is not None and not isinstance(p, int) else p
    J.bases, J.quals, J.off, J.ids, J.id_off, J.bc = adr(bk.bases), adr(bk.quals), adr(bk.off), adr(bk.ids), adr(bk.id_off), adr(bk.bc)
    J.cigar, J.desc, J.xa, J.sel_at = adr(cigar_ptr), adr(descs), adr(xas), adr(sel_at)
    J.names, J.name_off, J.rg, J.bx = adr(names_buf), name_off.ctypes.data, adr(rg_buf), adr(bx_buf)
    J.n_lines, J.cigar_lo = 2 * n_sel, cigar_lo
    J.has_rg, J.rg_len, J.bx_len, J.bc_len, J.is_haplotag = int(rg is not None), rg_len, len(opts.bx_index), opts.bc_len, opts.is_haplotag
    J.insert_min, J.insert_max = opts.insert_min, opts.insert_max
    n = L.emu_sam_format(C.byref(J), None, 0)
    buf = C.create_string_buffer(max(1, n) + 8)
    got = L.emu_sam_format(C.byref(J), buf, n)
    if got < 0:
        raise RuntimeError(f"emu_sam_format: code {got}")
    assert got == n
    return buf.raw[:n]


def sam_format_selection(sel, contig_names, opts) -> bytes:
    """A clouds.Selection made with opts.emit >= 1."""
    cig = C.cast(sel.b.cigar, C.c_void_p).value + 4 * sel.cigar_lo if sel.cigar_hi > sel.cigar_lo else None
    return sam_format(sel.bk, cig, sel.cigar_lo, sel.descs, sel.xas, sel.sel_at, sel.n_sel, contig_names, opts)
