"""The SAM formatter on the device (include/ema_sam.h: ema_sam_dev_*; csrc/k_sam.hip) on an MI355X against the oracle's restatement of
print_sam_record (oracle/sam.c; and the host formatter) on the same lines: the random cases of tests/sam_device_cases.py (every combination of aligned / unaligned record and mate, both strands, XA,
names from empty to 149 bytes, reads from 1 to 255 bases, 10x and haplotag barcodes, RG present / empty / absent), in one stretch and
in many (the two text buffers taking turns), and the bad-base flag.  The golden SAM cases and the stream tests go through the same
kernels by default (ema_stream_sam); tests/test_gpu_sam.py compares both formatters on a streamed run."""
import pytest

from ema_amd import sam
from sam_device_cases import CHROMS, Case

pytestmark = pytest.mark.gpu


def dev_text(fmt, case, so):
    return fmt.format(case.bk, case.cigar_ptr(), case.cigar_lo, case.cigar_hi, case.descs.ctypes.data, len(case.descs), case.xas.ctypes.data, case.n_xas,
                      case.sel_at.ctypes.data, case.n_sel, so)


@pytest.mark.parametrize("haplotag", [False, True])
def test_device_formatter_equals_the_host_formatter(haplotag, tuning):
    fmt = sam.DevFormatter(CHROMS)
    for seed, n_pairs, rg in ((3, 700, b"rg1\tSM:sample1"), (4, 64, None), (5, 33, b""), (6, 1, b"x")):
        case = Case(seed=seed, n_pairs=n_pairs, haplotag=haplotag)
        so = case.opts(rg=rg, bx=b"1" if seed != 4 else b"77")
        want = case.oracle_text(so)      # oracle/sam.c: print_sam_record restated call by call
        assert case.host_text(so) == want
        assert dev_text(fmt, case, so) == want
        tuning(sam_stretch_lines=128)      # the same lines in stretches of 128: buffers reused, the text written piecewise
        assert dev_text(fmt, case, so) == want
        tuning()
    fmt.close()


def test_device_formatter_flags_a_base_without_a_complement(tuning):
    fmt = sam.DevFormatter(CHROMS)
    case = Case(seed=5, n_pairs=40, haplotag=False, bases=b"ACGTNx")
    with pytest.raises(RuntimeError, match="-7"):
        dev_text(fmt, case, case.opts())
    good = Case(seed=9, n_pairs=40, haplotag=False)      # ... and the flag does not stick
    assert dev_text(fmt, good, good.opts()) == good.host_text(good.opts())
    fmt.close()


def test_an_empty_selection_writes_nothing():
    fmt = sam.DevFormatter(CHROMS)
    case = Case(seed=1, n_pairs=2, haplotag=False)
    assert fmt.format(case.bk, None, 0, 0, None, 0, None, 0, None, 0, case.opts()) == b""
    fmt.close()
