"""Rate of the SAM record formatter (dev aid; numbers quoted in DESIGN.md): ema_sam_format on the host's cores against
the oracle's stdio restatement of the reference's print_sam_record on the same lines."""
import os, sys, time, argparse, random, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import __graft_entry__
__graft_entry__.ensure_built()
from ema_amd import sam
import test_sam_format as T
ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=100000)
ap.add_argument("--copies", type=int, default=10, help="the line array is the generated one repeated this many times")
a = ap.parse_args()
arr0, n0, pool = T.make_lines(random.Random(1), a.pairs, False)
n = n0 * a.copies
arr = (sam.SamLine * n)()
for k in range(a.copies):
    C.memmove(C.byref(arr, k * n0 * C.sizeof(sam.SamLine)), arr0, n0 * C.sizeof(sam.SamLine))
o = sam.default_opts()
L = sam._lib()
ts = []
for _ in range(5):
    text, size = C.c_void_p(), C.c_size_t()
    t = time.perf_counter(); rc = L.ema_sam_format(arr, n, C.byref(o), C.byref(text), C.byref(size)); ts.append(time.perf_counter() - t)
    assert rc == 0
    nbytes = size.value
    L.ema_sam_free(text)
t_prod = sorted(ts)[2]
devnull = os.open("/dev/null", os.O_WRONLY)
tw = []
for _ in range(5):
    size = C.c_size_t()
    t = time.perf_counter(); rc = L.ema_sam_write(devnull, arr, n, C.byref(o), C.byref(size)); tw.append(time.perf_counter() - t)
    assert rc == 0 and size.value == nbytes
t_write = sorted(tw)[2]
t = time.perf_counter(); want = T.oracle_text(arr0, n0, o); t_orc = (time.perf_counter() - t) * a.copies
assert sam.format_lines(arr0, n0, o) == want
print(f"{n} lines, {nbytes / 1e6:.0f} MB of SAM text", flush=True)
print(f"ema_sam_format (median of 5): {t_prod * 1e3:.0f} ms = {n / t_prod / 1e6:.2f} M lines/s, {nbytes / t_prod / 1e9:.2f} GB/s "
      f"on {min(32, os.cpu_count())} host threads", flush=True)
print(f"ema_sam_write to /dev/null (no joining of the pieces): {t_write * 1e3:.0f} ms = {n / t_write / 1e6:.2f} M lines/s, "
      f"{nbytes / t_write / 1e9:.2f} GB/s", flush=True)
print(f"oracle (the reference's way: one thread, stdio call by call): {n / t_orc / 1e6:.2f} M lines/s, {nbytes / t_orc / 1e9:.2f} GB/s; "
      f"ratio {t_orc / t_prod:.1f}x", flush=True)
