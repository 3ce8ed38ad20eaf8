"""Host mirror of the SAM record formatter (include/ema_sam.h): the reference's print_sam_record
(reference src/samrecord.c:104-284) for a batch of lines in one call.  ctypes over the C ABI in libema_engine.so; no
fallback -- a missing library raises."""
from __future__ import annotations

import ctypes as C

from . import engine as _engine


class SamAlt(C.Structure):
    _fields_ = [("chrom", C.c_char_p), ("pos", C.c_uint32), ("edit_dist", C.c_int32), ("rev", C.c_int32), ("n_cigar", C.c_int32),
                ("cigar", C.POINTER(C.c_uint32))]


class SamRec(C.Structure):
    _fields_ = [("ident", C.c_char_p), ("chrom", C.c_char_p), ("chrom_id", C.c_uint32), ("pos", C.c_uint32),
                ("mapq", C.c_int32), ("score_mapq", C.c_int32), ("gamma", C.c_double),
                ("mate", C.c_uint8), ("rev", C.c_uint8), ("duplicate", C.c_uint8), ("pad_", C.c_uint8),
                ("cloud_id", C.c_int32), ("cloud_bad", C.c_int32), ("bc", C.c_uint64),
                ("read", C.c_char_p), ("qual", C.c_char_p), ("read_len", C.c_int32), ("mate_read_len", C.c_int32),
                ("mate_read", C.c_char_p), ("mate_qual", C.c_char_p), ("aln_pos", C.c_int64),
                ("aln_rev", C.c_int32), ("edit_dist", C.c_int32), ("n_cigar", C.c_int32), ("pad2_", C.c_int32),
                ("cigar", C.POINTER(C.c_uint32)), ("alts", C.POINTER(SamAlt)), ("n_alts", C.c_size_t)]


class SamLine(C.Structure):
    _fields_ = [("rec", C.POINTER(SamRec)), ("mate", C.POINTER(SamRec))]


class SamOpts(C.Structure):
    _fields_ = [("rg_id", C.c_char_p), ("bx_index", C.c_char_p), ("bc_len", C.c_int32), ("is_haplotag", C.c_int32),
                ("insert_min", C.c_int32), ("insert_max", C.c_int32)]


def _lib():
    L = _engine.load_library()
    if not getattr(L, "_sam_bound", False):
        L.ema_sam_opts_default.argtypes = [C.POINTER(SamOpts)]
        L.ema_sam_format.argtypes = [C.POINTER(SamLine), C.c_size_t, C.POINTER(SamOpts), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ema_sam_free.argtypes = [C.c_void_p]
        L.ema_sam_write.argtypes = [C.c_int, C.POINTER(SamLine), C.c_size_t, C.POINTER(SamOpts), C.POINTER(C.c_size_t)]
        L.ema_sam_header.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_char_p, C.c_int,
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L._sam_bound = True
    return L


def default_opts() -> SamOpts:
    o = SamOpts()
    _lib().ema_sam_opts_default(C.byref(o))
    return o


def format_lines(lines, n: int, opts: SamOpts) -> bytes:
    """lines: a ctypes array of SamLine (the caller keeps everything it points at alive).  Returns the SAM text."""
    L = _lib()
    text, size = C.c_void_p(), C.c_size_t()
    rc = L.ema_sam_format(lines, n, C.byref(opts), C.byref(text), C.byref(size))
    if rc != 0:
        raise RuntimeError(f"ema_sam_format failed (code {rc})")
    try:
        return C.string_at(text, size.value)
    finally:
        L.ema_sam_free(text)


def write_lines(fd: int, lines, n: int, opts: SamOpts) -> int:
    """The same text straight to an open file descriptor; returns the number of bytes written."""
    size = C.c_size_t()
    rc = _lib().ema_sam_write(fd, lines, n, C.byref(opts), C.byref(size))
    if rc != 0:
        raise RuntimeError(f"ema_sam_write failed (code {rc})")
    return size.value


def header(contigs, rg_line, version: bytes, argv) -> bytes:
    """write_sam_header's text: contigs = [(name, length)], rg_line = the whole @RG line or None, argv = the command line."""
    L = _lib()
    names = (C.c_char_p * max(1, len(contigs)))(*[n for n, _ in contigs])
    lens = (C.c_int32 * max(1, len(contigs)))(*[l for _, l in contigs])
    av = (C.c_char_p * len(argv))(*argv)
    text, size = C.c_void_p(), C.c_size_t()
    rc = L.ema_sam_header(names, lens, len(contigs), rg_line, version, len(argv), av, C.byref(text), C.byref(size))
    if rc != 0:
        raise RuntimeError(f"ema_sam_header failed (code {rc})")
    try:
        return C.string_at(text, size.value)
    finally:
        L.ema_sam_free(text)


# ---- the formatter on the device (ema_sam_dev_*, kernels in csrc/k_sam.hip): the same lines from the compact records ----
import numpy as _np

DESC_DTYPE = _np.dtype([("pair", "<u4"), ("rid", "<i4"), ("pos", "<u4"), ("cigar_off", "<u4"), ("n_cigar", "<i4"), ("edit_dist", "<i4"),
                        ("cloud_id", "<i4"), ("xa", "<i4"), ("mate", "u1"), ("rev", "u1"), ("duplicate", "u1"), ("cloud_bad", "u1"),
                        ("mapq", "u1"), ("gamma_len", "u1"), ("has_mate", "u1"), ("pad_", "u1"), ("gamma", "S12")])      # ema_sam_desc
XA_DTYPE = _np.dtype([("rid", "<i4"), ("pos", "<u4"), ("cigar_off", "<u4"), ("n_cigar", "<i4"), ("edit_dist", "<i4"), ("rev", "<i4")])      # ema_sam_xa
assert DESC_DTYPE.itemsize == 52 and XA_DTYPE.itemsize == 24


class DevFormatter:
    """ema_sam_dev_open / _format / _close: the formatter's kernels on `device`.  No fallback: without a GPU the open fails."""

    def __init__(self, contig_names, device: int = 0):
        L = _lib()
        L.ema_sam_dev_open.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int32, C.POINTER(C.c_void_p)]
        L.ema_sam_dev_close.argtypes = [C.c_void_p]
        L.ema_sam_dev_format.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_size_t, C.POINTER(SamOpts), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ema_sam_dev_last_error.restype = C.c_char_p
        names = (C.c_char_p * max(1, len(contig_names)))(*[n if isinstance(n, bytes) else n.encode() for n in contig_names])
        self._h = C.c_void_p()
        rc = L.ema_sam_dev_open(device, names, len(contig_names), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"ema_sam_dev_open failed ({rc}): {L.ema_sam_dev_last_error().decode()}")

    def format(self, bucket_struct, cigar_ptr, cigar_lo, cigar_hi, descs, n_descs, xas, n_xas, sel_at, n_sel, opts: SamOpts) -> bytes:
        """bucket_struct: a ctypes ema_bucket; cigar_ptr: operation cigar_lo of the batch's array; the rest as ema_clouds_out carries it."""
        L = _lib()
        text, size = C.c_void_p(), C.c_size_t()
        rc = L.ema_sam_dev_format(self._h, C.byref(bucket_struct), cigar_ptr, cigar_lo, cigar_hi, descs, n_descs, xas, n_xas, sel_at, n_sel,
                                  C.byref(opts), C.byref(text), C.byref(size))
        if rc != 0:
            raise RuntimeError(f"ema_sam_dev_format failed (code {rc}): {L.ema_sam_dev_last_error().decode()}")
        try:
            return C.string_at(text, size.value)
        finally:
            L.ema_sam_free(text)

    def format_selection(self, sel, opts: SamOpts) -> bytes:
        """A clouds.Selection made with opts.emit >= 1."""
        cig = C.cast(sel.b.cigar, C.c_void_p).value + 4 * sel.cigar_lo if sel.cigar_hi > sel.cigar_lo else None
        return self.format(sel.bk, cig, sel.cigar_lo, sel.cigar_hi, sel.descs, sel.n_descs, sel.xas, sel.n_xas, sel.sel_at, sel.n_sel, opts)

    def close(self):
        if self._h:
            _lib().ema_sam_dev_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
