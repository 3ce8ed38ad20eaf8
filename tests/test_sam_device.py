"""The SAM formatter's kernels (csrc/k_sam.hip: one lane per line, a length pass, a prefix sum, a writing pass) under the host SIMT
interpreter against the oracle's restatement of print_sam_record (and the host formatter) on the same lines -- every combination of aligned / unaligned record and mate, both strands
(reversed reads of every length from 1 up: the word-at-a-time copies and their byte-wise heads and tails), XA entries, positions
beyond 2^31, empty CIGARs, names from empty to 149 bytes, 10x and haplotag barcodes, RG present / empty / absent.  CPU only; the
GPU runs the same cases in tests/test_gpu_sam_device.py.  The interpreter is test infrastructure, not parity evidence."""
import pytest

import emu_lib
from sam_device_cases import CHROMS, Case


@pytest.mark.parametrize("haplotag", [False, True])
@pytest.mark.parametrize("rg", [b"rg1\tSM:sample1", b"", None, b"a-long-read-group-identifier.0123456789"])
def test_interpreter_formats_what_the_host_formatter_does(haplotag, rg):
    case = Case(seed=11 + int(haplotag), n_pairs=150, haplotag=haplotag)
    so = case.opts(rg=rg, bx=b"1" if rg != b"" else b"42")
    want = case.oracle_text(so)      # oracle/sam.c: print_sam_record restated call by call
    assert case.host_text(so) == want
    got = emu_lib.sam_format(case.bk, case.cigar_ptr(), case.cigar_lo, case.descs.ctypes.data, case.xas.ctypes.data, case.sel_at.ctypes.data,
                             case.n_sel, CHROMS, so)
    assert got == want
    assert got.count(b"\n") == 2 * case.n_sel


def test_a_base_without_a_complement_in_a_reversed_read_is_flagged():
    """print_sam_record's rc() asserts (reference src/samrecord.c:90-102); the host formatter returns EMA_EFORMAT, the kernel sets its flag."""
    case = Case(seed=5, n_pairs=40, haplotag=False, bases=b"ACGTNx")
    so = case.opts()
    with pytest.raises(RuntimeError):
        case.host_text(so)
    with pytest.raises(RuntimeError, match="-7"):
        emu_lib.sam_format(case.bk, case.cigar_ptr(), case.cigar_lo, case.descs.ctypes.data, case.xas.ctypes.data, case.sel_at.ctypes.data,
                           case.n_sel, CHROMS, so)
