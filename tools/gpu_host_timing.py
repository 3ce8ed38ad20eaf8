"""Dev aid: wall time of the host-facing calls (stage / run+sync / fetch / append) for one 1M-pair batch."""
import os, sys, time, argparse, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import bench, __graft_entry__
__graft_entry__.ensure_built()
a = argparse.Namespace(genome_mbp=0.0, pairs=int(sys.argv[1]) if len(sys.argv) > 1 else 1048576)
wd = os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid()); os.makedirs(wd, exist_ok=True)
prefix, pairs, _ = bench.build_workload(a, 0, 1, wd)
from ema_amd import engine as E
o = E.default_opts(); o.batch_pairs = a.pairs
eng = E.Engine(prefix, opts=o)
for rep in range(3):
    t0 = time.perf_counter(); eng.stage(pairs.bases, pairs.off); t1 = time.perf_counter()
    eng.run(); eng.sync(); t2 = time.perf_counter()
    L = eng._L; import ctypes as C
    p = C.POINTER(E.BatchOut)(); rc = L.ema_engine_fetch(eng._h, C.byref(p)); t3 = time.perf_counter()
    q = C.POINTER(E.AlnOut)()
    L.ema_batch_append_alignments.argtypes = [C.POINTER(E.BatchOut), C.c_void_p, C.POINTER(E.Opts), C.c_double, C.POINTER(C.POINTER(E.AlnOut))]
    rc2 = L.ema_batch_append_alignments(p, pairs.off.ctypes.data, C.byref(o), 0.001, C.byref(q)); t4 = time.perf_counter()
    L.ema_aln_free.argtypes = [C.POINTER(E.AlnOut)]; L.ema_aln_free(q); L.ema_batch_free(p)
    print(f"rep {rep}: stage {1e3*(t1-t0):.1f} ms, run+sync {1e3*(t2-t1):.1f} ms, fetch (C) {1e3*(t3-t2):.1f} ms rc={rc}, append {1e3*(t4-t3):.1f} ms rc={rc2}", flush=True)
