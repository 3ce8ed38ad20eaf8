"""Dev aid (CPU, oracle): the CPU baseline's thread scaling on the GPU box's host -- pairs/s, user and system seconds by thread
count, pinned to one socket, with glibc's default arena trimming and without (ORC_BENCH_DEFAULT_MALLOC=1 in a child process).
  python tools/cpu_scaling.py [N_PAIRS_PER_THREAD]      (uses the bench workdir's reference and first batch)"""
import glob, os, resource, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
if len(sys.argv) > 2 and sys.argv[2] == "child":
    import bench
    import oracle_lib as O
    from ema_amd import synth
    wd = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
    z = np.load(sorted(glob.glob(os.path.join(wd, "reads_*.npz")))[0])
    pairs = synth.Pairs(z["bases"], z["off"])
    per = int(sys.argv[1])
    cores, model = bench.one_socket_cpus()
    with bench.pinned_to(cores):
        idx, opt = O.Index(os.path.join(wd, "ref.fa")), O.default_opt()
        for th in (1, 8, 16, 32, len(cores)):
            n = min(pairs.n, per * th)
            s = pairs.subset(0, n)
            r0 = resource.getrusage(resource.RUSAGE_SELF)
            secs, _ = O.bench_pairs(idx, opt, s.bases, s.off, th)
            r1 = resource.getrusage(resource.RUSAGE_SELF)
            print(f"  {th:3d} threads: {n / secs:9.1f} pairs/s  wall {secs:6.2f}s user {r1.ru_utime - r0.ru_utime:7.1f}s sys {r1.ru_stime - r0.ru_stime:7.1f}s "
                  f"minor faults {r1.ru_minflt - r0.ru_minflt}", flush=True)
    sys.exit(0)
per = sys.argv[1] if len(sys.argv) > 1 else "3000"
for label, env in (("arenas keep their memory (mallopt in orc_bench_pairs)", {}), ("glibc defaults", {"ORC_BENCH_DEFAULT_MALLOC": "1"})):
    print(label, flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, os.path.abspath(__file__), per, "child"], env=e)
