#!/usr/bin/env python3
"""bench.py -- aligned read pairs/s of the `ema align` seed-and-extend hot path on MI355X.

One "step" = the whole hot path over one batch of 1 Mi synthetic read pairs: K1 seeding -> K2 chaining/extension -> K3
mate rescue -> K4 final alignment on the GPU, results packed and fetched to the host, and the reference's
append_alignments() stage (filters, MAPQ, likelihoods) on the host's cores -- i.e. everything the reference does per pair
in append_alignments() (reference src/align.c:986-1061: bwa_mem_mate_sw + one bwa_smith_waterman per candidate + the record
arithmetic), through the C ABI (include/ema_stream.h over include/ema_engine.h).  The K steps run over K DISTINCT batches
(10 x 1 Mi pairs = the 10 M pairs of BASELINE configs[1]) that are staged in HBM before the timed region starts, pipelined
over the engine's two sets of batch buffers as a caller streaming buckets would.

  python bench.py --gpus N --steps K --warmup W      (N > 1 without torch.distributed.run: the ranks are spawned here)

Rank 0 prints ONE JSON line.  `value` = pairs x steps / wall of that timed region (inputs resident, outputs on the host).
Beside it:
  boundary        : the same K batches from HOST buffers (nt4 conversion, 2-bit packing and H2D inside the timed region):
                    what a caller of the C ABI sees host to host (ema_stream_batches)
  engine_resident : kernels only, K steps queued back to back on one set of buffers, nothing fetched (round 1's `value`)
  roofline        : HBM roofline of K1, the kernel that moves the bytes (algorithmic bytes from the oracle's counters on
                    the same reads / its mean launch-series duration over the timed region, HIP events on its stream)
  roofline_k2b    : instruction-issue roofline of K2b, the largest kernel by time, from the committed PMC pass
  cpu_baseline    : the oracle (a CPU restatement, NOT upstream bwa: parity unpinned) on one socket's physical cores.
Multi-GPU: barcode buckets are independent: every rank runs its own K batches on its own replica of the index,
no data-path collective; the per-bucket statistics are all-gathered (RCCL) at the end.
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything initialises the GPU runtime: see ema_amd/csrc/engine.hip
import ctypes as C
import json
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
# VALU issue peak, MEASURED (tools/valu_issue_microbench.hip, profiles/r06_valu_issue_microbench.txt): full-rate 32-bit integer instructions
# (v_add_u32, v_and_b32) from >= 2 wavefronts per SIMD issue at one per 2.2-2.4 clocks per SIMD: 1,073 G wave-instructions/s chip-wide at
# 8 waves per SIMD (the guide's one-per-2-cycles figure would be 1,229 G at 2.4 GHz); ONE wavefront alone issues one per ~5 clocks, and
# v_bcnt_u32_b32 / v_lshlrev_b64 are half rate (593 G).  K2's stream is mostly full-rate 32-bit integer work.
VALU_PEAK_GINST = 1073.0
CHR20_LEN = 64_444_167
ENGINE_KNOBS = ("EMA_SEED_ROUNDS", "EMA_SEED_PARK", "EMA_SEED_BLOCKS_PER_CU", "EMA_FULL_SEED_LANE", "EMA_LANE_ALIGN", "EMA_FULL_OWN_STREAM",
                "EMA_KMER_K", "EMA_HEAVY_CHAINS", "EMA_SEED_TAIL", "EMA_SEED_LONG_WAVE", "EMA_LEAN_SEED_EXTENDS", "EMA_GRID", "EMA_LEAN_INTERVALS",
                "EMA_LEAN_REGIONS", "EMA_DEVICE_MERGE", "EMA_TUNING")      # (ema_amd/engine.py hands the EMA_<KNOB> ones to ema_engine_set_tuning)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpus_granted():
    """CPUs this job may really use: the affinity mask cut by the cgroup's CPU quota (the GPU boxes show a 64-core socket and
    grant a job 16-19 CPUs' worth of run time through cpu.max, r02b)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 8
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            t = open(path).read().split()
            if path.endswith("cpu.max"):
                if t[0] != "max":
                    n = min(n, max(1, int(float(t[0]) / float(t[1]) + 0.5)))
            else:
                q = int(t[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return n


def host_threads_for_rank(cpus, world):
    """Every rank stages, fetches and runs its append / cloud / formatter stages on host threads (default: up to 32 per rank);
    N ranks share the node's CPUs, so each takes its share and no more -- never more threads in total than CPUs granted
    (round 2 had a floor of 4 per rank: 8 ranks on 19 CPUs oversubscribed)."""
    return max(1, min(32, cpus // max(1, world)))


def pin_to_gpu_numa_node(local):
    """The rank's host threads on the NUMA node its GPU hangs off (sysfs: the GPU's PCI device -> numa_node -> cpulist), within
    the affinity mask the job already has.  Best effort: returns the node or None.  Must run before the engine starts threads."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus |= set(range(int(a), int(b or a) + 1))
        mine = cpus & os.sched_getaffinity(0)
        if mine:
            os.sched_setaffinity(0, mine)
            return node
    except Exception:      # noqa: BLE001 -- no sysfs entry, no permission: stay where we are
        pass
    return None


def spawn_ranks(n):
    """--gpus N > 1 outside torch.distributed.run: start the N ranks as child processes (nothing has touched the GPU in
    this process) and leave with their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd))


def genome_spec(args):
    if args.genome_mbp > 0:
        n_ctg = max(1, int(np.ceil(args.genome_mbp / 155.0)))          # chromosome-sized contigs (contig lengths are 32-bit)
        lens = [int(args.genome_mbp * 1e6 / n_ctg)] * n_ctg
        gname = f"synthetic {args.genome_mbp:g} Mbp, {n_ctg} contig{'s' if n_ctg > 1 else ''}"
    else:
        lens = [CHR20_LEN]
        gname = "synthetic chr20-scale (64,444,167 bp, 1 contig)"
    return lens, gname


def build_reference(args, workdir):
    """Rank 0 only: synthetic genome (genome.npy, reused if it is the same one) and its index.  Returns the stamp."""
    from ema_amd import synth, build_index
    lens, gname = genome_spec(args)
    prefix = os.path.join(workdir, "ref.fa")
    gpath = os.path.join(workdir, "genome.npy")
    stamp = json.dumps({"lens": lens, "seed": synth.GENOME_SEED, "builder": 3})
    t = time.time()
    try:
        have_genome = open(prefix + ".gstamp").read() == stamp and os.path.exists(gpath)
    except OSError:
        have_genome = False
    have_index = have_genome and all(os.path.exists(prefix + e) for e in (".bwt", ".sa", ".fsa", ".pac", ".ann", ".amb", ".stamp")) \
        and open(prefix + ".stamp").read() == stamp
    if have_index:
        log(f"[rank 0] genome {gname} and its index found in {workdir}: reused")
        return stamp
    if have_genome:
        flat = np.load(gpath, mmap_mode="r")
        ctg, at = [], 0
        for n in lens:
            ctg.append(np.asarray(flat[at:at + n])); at += n
    else:
        ctg = synth.make_genome_native(lens, seed=synth.GENOME_SEED)      # csrc/synth_genome.cpp: seconds, not the 45 s of numpy's generator
        for e in (".gstamp", ".stamp"):
            if os.path.exists(prefix + e):
                os.remove(prefix + e)
        for f in os.listdir(workdir):      # reads cached for another genome
            if f.startswith("reads_"):
                os.remove(os.path.join(workdir, f))
        np.save(gpath, np.concatenate(ctg))
        with open(prefix + ".gstamp", "w") as f:
            f.write(stamp)
    log(f"[rank 0] genome {gname}: {time.time() - t:.1f}s")
    t = time.time()
    if os.path.exists(prefix + ".stamp"):
        os.remove(prefix + ".stamp")
    synth.write_fasta(prefix, ctg, names=["chr20"] if len(ctg) == 1 else [f"chr{i + 1}" for i in range(len(ctg))])
    build_index(prefix)
    with open(prefix + ".stamp", "w") as f:
        f.write(stamp)
    log(f"[rank 0] index built in {time.time() - t:.1f}s")
    return stamp


def make_batches(args, rank, world, workdir, n_batches):
    """This rank's n_batches distinct batches of read pairs (generated by worker processes from the shared genome.npy;
    cached in workdir so that the driver's back-to-back runs and the profiler passes reuse them)."""
    from ema_amd import synth
    lens, _ = genome_spec(args)
    gpath = os.path.join(workdir, "genome.npy")
    t = time.time()
    todo, paths = [], []
    for k in range(n_batches):
        seed = synth.READS_SEED + 7919 * rank + 104729 * k
        path = os.path.join(workdir, f"reads_g{args.genome_mbp:g}_n{args.pairs}_s{seed}.npz")
        paths.append(path)
        if not os.path.exists(path):
            todo.append((gpath, lens, args.pairs, seed, 127, 150, path))
    if todo:
        import multiprocessing as mp
        n_proc = max(1, min(len(todo), (os.cpu_count() or 8) // max(1, world)))
        with mp.get_context("spawn").Pool(n_proc) as pool:      # spawn: this process may already hold the GPU
            pool.map(synth.bench_batch, todo)
    out = []
    for path in paths:
        z = np.load(path)
        out.append(synth.Pairs(z["bases"], z["off"], z["barcodes"] if "barcodes" in z.files else None))
    log(f"[rank {rank}] {n_batches} batches x {args.pairs} pairs: {len(todo)} simulated, {n_batches - len(todo)} from {workdir}, {time.time() - t:.1f}s")
    return out


def algorithmic_bytes(stats, sa_width):
    """bytes the algorithm must move per unit, from the oracle's counters on the same reads.
    K1 (seeding): two 32-byte rank blocks per bwt_extend + the read.  K2..K4: SA rows, reference windows (2 bit/base),
    region records, CIGAR ops."""
    k1 = 64 * stats["n_ext"] + stats["l_read"]
    rest = sa_width * stats["n_occ"] + stats["w_ref"] // 4 + 88 * stats["n_regs"] + 4 * stats["n_cigar"]
    return k1, rest


def one_socket_cpus():
    """One hardware thread per physical core of NUMA node 0 (the socket the baseline is pinned to), and the CPU model."""
    def parse(s):
        out = []
        for part in s.strip().split(","):
            if "-" in part:
                a, b = part.split("-"); out.extend(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out
    allowed = os.sched_getaffinity(0)
    try:
        node0 = [c for c in parse(open("/sys/devices/system/node/node0/cpulist").read()) if c in allowed]
    except OSError:
        node0 = sorted(allowed)
    cores, seen = [], set()
    for c in node0:
        try:
            sib = tuple(parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
        except OSError:
            sib = (c,)
        if sib not in seen:
            seen.add(sib); cores.append(c)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip(); break
    except OSError:
        pass
    return cores or sorted(allowed), model


class pinned_to:
    """Every thread of this process (OpenMP pool threads included) on the given CPUs for the duration; restored afterwards."""

    def __init__(self, cpus):
        self.cpus, self.saved = set(cpus), {}

    def __enter__(self):
        for tid in os.listdir("/proc/self/task"):
            try:
                self.saved[int(tid)] = os.sched_getaffinity(int(tid))
                os.sched_setaffinity(int(tid), self.cpus)
            except OSError:
                pass
        return self

    def __exit__(self, *a):
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), self.saved.get(int(tid), self.saved.get(os.getpid(), self.cpus)))
            except OSError:
                pass


def pmc_table(name):
    """profiles/<name>: kernel,counter,launches,mean_per_launch,min,max,sum_over_run (tools/pmc_summary.py --csv)."""
    import csv
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    t = {}
    for r in csv.DictReader(open(path)):
        t[(r["kernel"], r["counter"])] = (float(r["sum_over_run"]), float(r["launches"]))
    return t


def sam_leg(args, eng, batches, workdir, world):
    """SURVEY 8(d)'s end-to-end figure beside the hot path's, on BASELINE configs[2]'s shape: 500 barcode buckets streamed through ONE
    ema_stream_sam call (reader, engine, append stage, clouds / EM / duplicates, formatter) to /dev/null, `-d` on -- every bucket its own
    `ema align -s` process (cloud numbers from 0, its own stream of -d draws: ema_cloud_opts.seed_private) -- and the same call without -d.
    The buckets are cut from the bench batches (50 distinct files of a tenth of a 512 Ki-pair half-batch each, taken ten times round:
    26.2 M pairs per call at the default scale).  Runs last: it stages its own input over the resident batches.  N = 1 only."""
    if world != 1 or any(b.barcodes is None for b in batches[:2]):
        return None
    from ema_amd import shard, stream, synth
    try:
        n_buckets = 500
        per = max(256, min(52428, args.pairs // 20))      # 26.2 M pairs over 500 buckets at the default scale (configs[2]: 100 M reads in 500 buckets = 100,000 pairs each)
        n_files = min(50, 5 * len(batches))
        t = time.time()
        paths = []
        for k in range(n_files):
            src = batches[k % len(batches)]
            lo = (k // len(batches)) * per
            path = os.path.join(workdir, f"bench-bucket-{k:03d}")
            synth.write_special_fastq_fixed(path, src.subset(lo, lo + per))
            paths.append(path)
        t_write = time.time() - t
        order = [paths[k % n_files] for k in range(n_buckets)]
        tot = n_buckets * per
        fd = os.open("/dev/null", os.O_WRONLY)
        stream.stream_sam(eng, paths[:2], fd, rg_id=b"rg1")      # warm-up: page cache, buffers
        import resource

        def one(density):
            stream.host_cpu_seconds(reset=True)
            ru0 = resource.getrusage(resource.RUSAGE_SELF)
            t0 = time.perf_counter()
            bst, sst = stream.stream_sam(eng, order, fd, rg_id=b"rg1", density_opt=density, density_seed=1500000000 if density else None)
            dt = time.perf_counter() - t0
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            cpu_s = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
            by_stage = stream.host_cpu_seconds()
            # the per-bucket records a rank would gather (ema_amd/shard.py: every field of ema_bucket_stats and ema_sam_stats), summed here
            table = shard.gather_stats(np.array([shard.bucket_stats(b, q) for b, q in zip(bst, sst)], dtype=np.int64), n_buckets)
            tsum = shard.stats_as_dict(table.sum(axis=0))
            return {"value": round(tot / dt, 1), "unit": "pairs/s", "seconds": round(dt, 3), "density_optimiser": bool(density),
                    "stage_seconds": {"reader": round(sum(x["read_s"] for x in bst), 3), "append": round(sum(x["append_s"] for x in bst), 3),
                                      "clouds_em_duplicates": round(sum(x["select_s"] for x in sst), 3), "formatter_and_write": round(sum(x["write_s"] for x in sst), 3)},
                    "host_cpu_seconds_per_million_pairs": round(cpu_s / tot * 1e6, 3), "host_cpus_busy": round(cpu_s / dt, 1),
                    "host_cpu_seconds_per_million_pairs_by_stage": {k: round(v / tot * 1e6, 3) for k, v in by_stage.items()},
                    "bucket_records": {"buckets": int(table.shape[0]), "fields_per_bucket": int(table.shape[1]),
                                       "totals": {k: tsum[k] for k in ("pairs", "records", "redone_pairs", "sam_lines", "sam_mapped", "sam_proper", "sam_duplicates",
                                                                        "sam_with_xa", "sam_clouds", "sam_bad_clouds", "sam_mapq_hist")}}}
        plain = one(False)
        with_d = one(True)
        os.close(fd)
        for path in paths:
            os.remove(path)
        log(f"[rank 0] bucket files -> SAM text, {n_buckets} buckets of {per} pairs: {with_d['value']:.0f} pairs/s with -d, {plain['value']:.0f} without (files written in {t_write:.1f}s)")
        out = dict(with_d)
        out.update({"buckets": n_buckets, "pairs_per_bucket": per, "distinct_bucket_files": n_files,
                    "what": "BASELINE configs[2]'s shape: 500 bucket files (preproc's one-pair-per-line form, page cache) -> SAM text on /dev/null through ONE "
                            "ema_stream_sam call with -d on: reader (parse, sort by barcode and gather on the device; the reads stay in HBM), staging, K1-K4, fetch, "
                            "append stage, clouds / EM / duplicate marking / density optimiser on the host's threads (r06: every bucket its own reference "
                            "process -- cloud numbers from 0, its own glibc random_r stream seeded seed + k -- so up to three buckets are in the cloud stage at a time), "
                            "formatter on the device; small buckets share passes; host stages on the CPUs the box grants",
                    "without_density_optimiser": plain})
        return out
    except Exception as e:      # an extra: never at the cost of the line
        log(f"[rank 0] bucket files -> SAM text leg failed: {e}")
        return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="default: 10 steps x 1 Mi pairs = the 10 M pairs of BASELINE configs[1]")
    ap.add_argument("--warmup", type=int, default=3, help="untimed passes before the timed ones; three, so that the pipeline's pooled page-locked batches (one per pass in flight) exist before the clock starts")
    ap.add_argument("--pairs", type=int, default=1048576, help="pairs per step and per GPU (one batch)")
    ap.add_argument("--batches", type=int, default=0, help="distinct batches resident per GPU (default min(steps, 10); steps cycle through them)")
    ap.add_argument("--genome-mbp", type=float, default=3100.0,
                    help="size of the synthetic reference; default GRCh38-scale (3.1 Gbp in 20 contigs, BASELINE configs[1]); "
                         "0 = chr20-scale (64.4 Mbp, one contig: configs[0]'s reference)")
    ap.add_argument("--cpu-sample", type=int, default=1000000, help="pairs of the same workload timed on the host CPU (one socket)")
    ap.add_argument("--streams", type=int, default=0, help="slices of a batch on their own HIP streams (0 = engine default)")
    ap.add_argument("--lean-seed-extends", type=int, default=int(os.environ.get("EMA_LEAN_SEED_EXTENDS", "0")),
                    help="engine option lean_seed_extends (0 = engine default; EMA_LEAN_SEED_EXTENDS sets the default for A/B runs)")
    ap.add_argument("--two-sets", action="store_true", help="the older schedule: alternate batches on two sets of batch buffers, one pass each "
                                                            "(default: one set, passes queued up to three deep with the layout and packing on the device)")
    ap.add_argument("--spot-check", type=int, default=24000, help="regular pairs of the timed steps re-aligned by the oracle afterwards (besides "
                                                                  "every pair the full-capacity tier redid in two of the steps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-at-world-1", action="store_true", help="run the distributed control flow (torch.distributed process group, barriers, "
                                                                    "all-reduces, statistics all-gather) even at world size 1")
    ap.add_argument("--no-extras", action="store_true", help="skip the boundary / engine_resident / isolated passes (profiling runs)")
    ap.add_argument("--no-sam-leg", action="store_true", help="skip the bucket files -> SAM text leg (A/B runs of engine knobs)")
    ap.add_argument("--allow-capacity-flags", action="store_true", help="print the (invalid) line even if reads overflowed an engine capacity")
    ap.add_argument("--engine-module", default="ema_amd", help=argparse.SUPPRESS)      # tests/test_bench_control_flow.py: a stand-in engine
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        log(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus} (or plain `python bench.py --gpus N`)")
        raise SystemExit(2)
    node_cpus = cpus_granted()
    # --dist-at-world-1: the distributed control flow (process group, barriers, agree() all-reduces, the statistics all-gather on the
    # device) also at world size 1 -- what a one-GPU box can run of it on hardware (tests/test_gpu_bench_dist.py)
    use_dist = world > 1 or args.dist_at_world_1
    if args.dist_at_world_1 and "RANK" not in os.environ:      # (without a launcher: a rendezvous of one on this host)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            free_port = sk.getsockname()[1]
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port))
    if world > 1 and "EMA_HOST_THREADS" not in os.environ:
        os.environ["EMA_HOST_THREADS"] = str(host_threads_for_rank(node_cpus, world))
    dist = None
    # RCCL ("nccl") on the GPU box; EMA_BENCH_BACKEND=gloo runs the same control flow on CPU tensors (the world-size-2 test)
    backend = os.environ.get("EMA_BENCH_BACKEND", "nccl")
    tdev = "cuda" if backend == "nccl" else "cpu"
    if use_dist:
        import torch
        import torch.distributed as dist
        import datetime
        # rank 0 builds genome and index (minutes at the default scale) while the others wait at a barrier
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            torch.cuda.set_device(local)
            numa = pin_to_gpu_numa_node(local)
            log(f"[rank {rank}] host threads: {os.environ.get('EMA_HOST_THREADS')} of {node_cpus} CPUs granted to the node's {world} ranks"
                + (f", pinned to NUMA node {numa} (the GPU's)" if numa is not None else ""))
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=30))
            log(f"[rank {rank}] process group: backend {dist.get_backend()} (RCCL), world size {dist.get_world_size()}, device cuda:{local}")
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=30))

    def agree(value, op="min"):
        """rank 0's float for everybody (a plain tensor collective, like the ones below)"""
        if not use_dist:
            return value
        g = torch.tensor([value], dtype=torch.float64, device=tdev)
        dist.all_reduce(g, op=dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.MAX)
        return float(g.item())

    import __graft_entry__
    if rank == 0:
        __graft_entry__.ensure_built()
    if use_dist:
        dist.barrier()

    import tempfile
    workdir = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
    os.makedirs(workdir, exist_ok=True)
    # A reference of G bases needs ~21 G bytes of files (flat suffix array: 16 G) and ~26 G bytes of host memory while the
    # index is built.  A box that cannot hold the requested reference gets the chr20-scale one instead, and the JSON says so
    # -- better a line on the smaller workload, labelled as such, than none.
    fallback = ""
    failed = 0.0
    if rank == 0 and args.genome_mbp > 0:
        from ema_amd import synth
        lens, _ = genome_spec(args)
        want = json.dumps({"lens": lens, "seed": synth.GENOME_SEED, "builder": 3})
        try:
            have = open(os.path.join(workdir, "ref.fa.stamp")).read() == want      # the index of THIS genome is already there
        except OSError:
            have = False
        if not have:
            import shutil
            need_disk, need_ram = 21e6 * args.genome_mbp, 26e6 * args.genome_mbp
            free_disk = shutil.disk_usage(workdir).free
            free_ram = need_ram
            try:      # MemAvailable counts the page cache that can be dropped
                for line in open("/proc/meminfo"):
                    if line.startswith("MemAvailable:"):
                        free_ram = int(line.split()[1]) * 1024
            except OSError:
                pass
            if free_disk < need_disk or free_ram < need_ram:
                fallback = (f"FALLBACK to the chr20-scale reference: {args.genome_mbp:g} Mbp needs {need_disk / 1e9:.0f} GB of disk in {workdir} "
                            f"({free_disk / 1e9:.0f} GB free) and {need_ram / 1e9:.0f} GB of host memory ({free_ram / 1e9:.0f} GB free); ")
                log(f"[rank {rank}] {fallback}")
                args.genome_mbp = 0.0
    args.genome_mbp = agree(args.genome_mbp if rank == 0 else 1e18)
    if rank == 0:
        try:
            build_reference(args, workdir)
        except BaseException as e:      # noqa: BLE001 -- the other ranks must not wait out the collective timeout
            log(f"[rank 0] building the reference failed: {e!r}")
            failed = 1.0
    if agree(failed, "max") > 0:
        if use_dist:
            dist.destroy_process_group()
        raise SystemExit(3)
    lens, gname = genome_spec(args)
    prefix = os.path.join(workdir, "ref.fa")
    n_batches = args.batches if args.batches > 0 else min(max(1, args.steps), 10)
    if args.two_sets:
        n_batches += n_batches & 1      # alternate batches on alternate sets of batch buffers: the same number on each
    batches = make_batches(args, rank, world, workdir, n_batches)
    if rank == 0:
        # The reference, its index (60 GB at the default scale) and the batches may just have been WRITTEN: let the kernel finish writing
        # them back before anything is timed -- the timed region's host side (page-locked downloads, the append stage's threads) shares
        # the box's memory system with that I/O, and the first run on a fresh box was the slow one in three of four calls of round 5
        # (150-158 ms per step against 142-146 for the same kernels, `profiles/r05_ab.txt`).  Outside every timed region.
        t_sync = time.time()
        os.sync()
        log(f"[rank 0] sync of the files written before the timed legs: {time.time() - t_sync:.1f}s")

    import importlib
    stream = importlib.import_module(args.engine_module + ".stream")
    engine_mod = importlib.import_module(args.engine_module + ".engine")
    Engine, default_opts = engine_mod.Engine, engine_mod.default_opts
    o = default_opts()
    o.batch_pairs = args.pairs
    o.n_streams = args.streams
    o.lean_seed_extends = args.lean_seed_extends
    stock, stock_rows = {}, None
    if not args.no_extras and world == 1:
        # The same index as `bwa index` leaves it (no flat suffix array file): an engine of its own (small batch geometry) whose suffix array
        # is bwa's sampled .sa expanded on the device.  FIRST, while the HBM is empty, and closed again before the bench's engine opens:
        # two index replicas and the bench's batch buffers do not fit 288 GB together.
        try:
            sdir = os.path.join(workdir, "stock_index")
            os.makedirs(sdir, exist_ok=True)
            for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
                dst = os.path.join(sdir, "ref.fa" + ext)
                if not os.path.lexists(dst):
                    os.symlink(prefix + ext, dst)
            o2 = default_opts()
            o2.batch_pairs = 16384
            t = time.time()
            e2 = Engine(os.path.join(sdir, "ref.fa"), device=local, opts=o2)
            stock["engine_open_stock_bwa_index_s"] = round(time.time() - t, 2)
            stock_rows = e2.debug_sa(12345, 4096)
            e2.close()
            del e2
            log(f"[rank 0] engine open on the index without its .fsa (sampled .sa expanded on the device): {stock['engine_open_stock_bwa_index_s']}s")
        except Exception as e:      # noqa: BLE001 -- an extra: never at the cost of the line
            log(f"[rank 0] stock-index open failed: {e}")
    t = time.time()
    eng = Engine(prefix, device=local, opts=o)
    open_s = time.time() - t
    log(f"[rank {rank}] engine open (index in HBM) {open_s:.1f}s")
    if stock_rows is not None:
        stock["stock_bwa_index_rows_equal_the_flat_file"] = bool((stock_rows == eng.debug_sa(12345, 4096)).all())
    so = stream.default_opts()
    so.n_engines = 2 if args.two_sets else 1
    peer = eng.peer() if args.two_sets else None
    n_sets = 2 if peer is not None else 1
    if n_sets == 1 and args.two_sets:
        log(f"[rank {rank}] no device memory for a second set of batch buffers: one set")
        so.n_engines = 1
    slots = (n_batches + n_sets - 1) // n_sets
    t = time.time()
    for k, p in enumerate(batches):          # nt4 conversion + packing + H2D of every batch: outside the timed region
        (eng if k % n_sets == 0 else peer).stage_slot(k // n_sets, p.bases, p.off)
    log(f"[rank {rank}] {n_batches} batches staged in HBM ({n_sets} set{'s' if n_sets > 1 else ''} of batch buffers) {time.time() - t:.1f}s")

    def sync_all():
        eng.sync()
        if peer is not None:
            peer.sync()
        if use_dist:
            if tdev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    # parity spot check inside the run: candidates of sampled pairs of EVERY timed step -- regular ones and pairs that went
    # through the full-capacity tier -- are copied by the sink (vectorised: ema_amd.engine.gather_pairs) and compared with the
    # oracle afterwards, digest against digest (oracle/pair.c, orc_digest_pairs, on all granted threads); a mismatch is fatal.
    # Two of the timed steps are checked in depth: EVERY pair the full-capacity tier redid and thousands of regular ones.
    rng = np.random.default_rng(12345 + rank)
    kept = []
    # [r5] The in-depth part of the check -- EVERY pair the full-capacity tier redid and thousands of regular ones, of two batches --
    # is taken from two EXTRA passes over batches 0 and 1 right after the clock stops (same engine, same path, same input slots):
    # inside the timed region it was ~0.1 s of the checker's copying in the stream's delivery thread, i.e. up to 5 ms per step at 20
    # steps of measurement overhead in `value`.  Every TIMED step is still sampled (48 pairs: 24 regular + 24 of the full tier's).
    deep = {"steps": set()}
    n_deep = 2 if args.steps >= 2 else 1
    deep_regular = max(24, args.spot_check // (2 * n_deep))

    def keep_sample(k, pb, n_reg, n_full):
        ob = pb.contents
        n = int(ob.n_pairs)
        nr = int(ob.n_redone)
        redone = np.ctypeslib.as_array(ob.redone, shape=(max(nr, 1),))[:nr].astype(np.int64)
        if k in deep["steps"]:
            n_reg, pick = deep_regular, redone
        else:
            pick = redone[rng.integers(0, nr, min(n_full, nr))] if nr else redone
        ids = np.unique(np.concatenate([rng.integers(0, n, min(n_reg, n)), pick]))
        cand, pool, read_off = engine_mod.gather_pairs(ob, ids)
        kept.append((k % n_batches, ids, cand, pool, read_off, len(np.intersect1d(ids, redone))))

    tallies = {"flags": 0}

    sink_cpu = [0.0]      # CPU seconds of the checker's own sampling inside the sink (not the product's: reported apart)

    step_done = []       # when the sink saw each step's results (timed region only): the spread shows a hiccup where it happened

    def make_sink(sample):
        def cb(_user, k, _pbk, pb, _pa):
            try:
                if sample and not deep["steps"]:
                    step_done.append(time.perf_counter())
                if sample:
                    t_ = time.thread_time()
                    keep_sample(int(k), pb, 24, 24)
                    sink_cpu[0] += time.thread_time() - t_
                return 0
            except BaseException as e:      # noqa: BLE001
                log(f"sink failed: {e!r}")
                return -100
        return stream.SINK(cb)

    offs = [batches[k % n_batches].off for k in range(max(args.steps, args.warmup))]
    cold_first_pass_s = None
    if args.warmup:
        # the first untimed pass alone, timed apart: it pays what no later pass pays (the pooled page-locked batches -- hipHostMalloc of
        # the first set -- the result sets' device allocations, cold caches); reported as `cold_first_pass_s`, never part of `value`
        t_c = time.perf_counter()
        stream.stream_resident(eng, offs[:1], slots, opts=so, raw_sink=make_sink(False))
        sync_all()
        cold_first_pass_s = time.perf_counter() - t_c
        if args.warmup > 1:
            stream.stream_resident(eng, offs[1:args.warmup], slots, opts=so, raw_sink=make_sink(False))
    sync_all()
    # ---- timed region: K steps = K batches (distinct up to n_batches), inputs resident, results + records on the host
    sink = make_sink(True)
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    st_timed = stream.stream_resident(eng, offs[:args.steps], slots, opts=so, raw_sink=sink)
    sync_all()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    n_kept_timed = len(kept)
    gaps = np.diff(np.array([t0] + step_done[:args.steps])) * 1e3
    step_gaps = None if len(gaps) < 2 else {"first": round(float(gaps[0]), 1), "median": round(float(np.median(gaps[1:])), 1),
                                            "max": round(float(gaps[1:].max()), 1), "max_at_step": int(np.argmax(gaps[1:])) + 1}
    deep["steps"] = set(range(n_deep))      # (the two passes below are sampled in depth)
    stream.stream_resident(eng, offs[:n_deep], slots, opts=so, raw_sink=make_sink(True))
    sync_all()
    deep["steps"] = set()
    host_cpu_s = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)      # this rank's host threads over the timed region
    host_cpu_s = max(0.0, host_cpu_s - sink_cpu[0])      # ... without the spot check's sampling in the sink (the checker, not the product)
    if use_dist:
        elapsed = agree(elapsed, "max")
        host_cpu_s = agree(host_cpu_s, "max")
    for s in st_timed:
        tallies["flags"] |= s["capacity_flags"]
    any_flag = int(agree(float(tallies["flags"] != 0), "max"))
    if any_flag:
        log(f"[rank {rank}] ERROR: reads exceeded an engine capacity (status bits {tallies['flags']}): their pairs have no candidates, "
            f"so the timed steps skipped work and the number is not valid")
        if not args.allow_capacity_flags:
            eng.close()
            if use_dist:
                dist.destroy_process_group()
            raise SystemExit(2)
    # mean launch durations over the timed region (HIP events on the launching streams, read after every step's sync)
    kernel_ms = {k: float(np.mean([s[k] for s in st_timed])) for k in ("seed_ms", "extend_ms", "rescue_ms", "final_ms", "full_tier_ms")}

    boundary = resident = None
    kernel_ms_isolated = None
    if not args.no_extras:
        # ---- kernels only, queued back to back on one set (nothing fetched): round 1's figure, for continuity
        for _ in range(args.warmup):
            eng.run_slot(0)
        sync_all()
        t2 = time.perf_counter()
        for k in range(args.steps):
            eng.run_slot(k % slots)      # the batches staged on this set
        sync_all()
        el_r = agree(time.perf_counter() - t2, "max") if use_dist else time.perf_counter() - t2
        resident = {"value": round(args.pairs * args.steps * world / el_r, 1), "unit": "pairs/s", "ms_per_step": round(el_r / args.steps * 1e3, 3),
                    "what": "K1-K4 only, steps queued back to back on one set of batch buffers, inputs resident, nothing fetched"}
        # ---- one pass on its own, from queueing to records on the host (what --sync-each-step used to show: nothing overlaps)
        sync_all()
        t3 = time.perf_counter()
        stream.stream_resident(eng, offs[:1], slots, opts=so, raw_sink=make_sink(False))
        sync_all()
        el_1 = agree(time.perf_counter() - t3, "max") if use_dist else time.perf_counter() - t3
        resident["single_pass"] = {"value": round(args.pairs * world / el_1, 1), "unit": "pairs/s", "ms": round(el_1 * 1e3, 3),
                                   "what": "one batch alone through the timed region's path: three lean slices, the full-capacity tier, pack, D2H, assembly, "
                                           "append stage with nothing to overlap; the timed region's per-step time approaches engine_resident's as steps grow"}
        eng.run(serial=True)      # one extra untimed pass with the slices one after another: launches in isolation
        eng.sync()
        tm = eng.timing()
        kernel_ms_isolated = {k: tm[k] for k in kernel_ms}
        log(f"[rank {rank}] full-capacity tier K1..K4 ms (isolated pass): {tm['full_ms']}")
        # ---- the same K batches from host buffers: stage (nt4 + packing + H2D) inside the timed region too.  LAST of the extras: it
        # stages into input slots that hold resident batches (the legs above run the batches the labels say)
        hb = [(batches[k % n_batches].bases, batches[k % n_batches].off) for k in range(args.steps)]
        sync_all()
        t1 = time.perf_counter()
        st_b = stream.stream_batches(eng, hb, opts=so, raw_sink=make_sink(False))
        sync_all()
        el_b = agree(time.perf_counter() - t1, "max") if use_dist else time.perf_counter() - t1
        boundary = {"value": round(args.pairs * args.steps * world / el_b, 1), "unit": "pairs/s", "ms_per_step": round(el_b / args.steps * 1e3, 3),
                    "what": "host buffers in (ASCII reads) -> candidates + append_alignments records in host memory, "
                            "ema_stream_batches over the same batches: nt4 conversion, 2-bit packing, H2D, K1-K4, pack, D2H, append stage, "
                            f"pipelined: " + ("two sets of batch buffers, one pass each" if n_sets == 2 else "one set, passes queued up to three deep (ema_engine_run_async), staging / kernels / fetch + append of consecutive batches overlapping"),
                    "host_s_per_step": {"align_call": round(float(np.mean([s["align_s"] for s in st_b])), 4),
                                        "append": round(float(np.mean([s["append_s"] for s in st_b])), 4)}}
    n_slices = eng.n_streams

    from ema_amd import shard
    local_stats = np.array([shard.bucket_stats(s) for s in st_timed], dtype=np.int64).sum(axis=0)[None, :]      # (the whole record of shard.STAT_FIELDS; its SAM part is the SAM leg's)
    # the "trivial RCCL gather of per-bucket statistics" of the north star: one record per rank, O(100 B) over xGMI
    gathered = shard.gather_stats(local_stats, world, device=(tdev if use_dist and tdev == "cuda" else None))

    bad = 0
    out = None
    if rank == 0:
        import oracle_lib as O
        idx, opt = O.Index(prefix), O.default_opt()
        t = time.time()
        n_checked = n_checked_full = n_checked_timed = 0
        for i_kept, (b, ids, cand, pool, read_off, n_full) in enumerate(kept):
            if i_kept < n_kept_timed:
                n_checked_timed += len(ids)
            sub = batches[b].take(ids)
            want, _secs = O.digest_pairs(idx, opt, sub.bases, sub.off, node_cpus)
            got = O.cand_digest(cand, pool, read_off)
            n_checked += len(ids); n_checked_full += n_full
            for r in np.nonzero(got != want)[0][:3 if bad < 3 else 0]:      # the first few in full
                p = int(ids[r >> 1]); m = int(r & 1)
                ref = O.align_pair(idx, opt, batches[b].read(2 * p), batches[b].read(2 * p + 1))[m]
                mine = cand[int(read_off[r]):int(read_off[r + 1])]
                log(f"MISMATCH batch {b} pair {p} mate {m + 1}: engine {[(int(c['rb']), int(c['re']), int(c['score']), int(c['pos']), int(c['NM'])) for c in mine[:3]]} "
                    f"oracle {[(d['rb'], d['re'], d['score'], d['pos'], d['NM']) for d in ref[:3]]}")
            bad += int((got != want).sum())
        log(f"[rank 0] oracle spot check: {n_checked} pairs ({n_checked_timed} of the timed steps, the others of two passes over batches 0 and 1 right after them), {n_checked_full} of them through the full-capacity tier "
            f"({bad} reads differ) {time.time() - t:.1f}s on {node_cpus} threads")
    bad = int(agree(float(bad), "max"))
    if bad:
        log(f"ERROR: {bad} reads of the spot check differ from the oracle: no bench line")
        eng.close()
        if use_dist:
            dist.destroy_process_group()
        raise SystemExit(2)

    if rank == 0:
        # algorithmic bytes: the oracle's counters on a sample of the SAME reads, scaled to the batch
        n_1 = min(8000, batches[0].n)
        s1 = batches[0].subset(0, n_1)
        O.stats_reset()
        t_1, _ = O.bench_pairs(idx, opt, s1.bases, s1.off, 1)
        st = O.stats_get()
        scale = args.pairs / float(n_1)
        sa_width = eng.index_info()["sa_width"]
        k1_bytes, rest_bytes = algorithmic_bytes(st, sa_width)
        k1_bytes *= scale / n_slices           # one launch series covers one slice of the batch
        rest_bytes *= scale / n_slices
        dom_ms = kernel_ms["seed_ms"]
        achieved = k1_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        knobs = [k for k in ENGINE_KNOBS if k in os.environ]
        default_run = (args.pairs == 1048576 and n_slices == 3 and args.lean_seed_extends == 0 and not knobs)
        traffic, traffic_source, traffic_stale = None, None, None
        pmc_name = next((n for n in ({0.0: ("r04_pmc_chr20.csv", "r03_pmc_chr20.csv"), 3100.0: ("r06_pmc_grch38scale.csv", "r05_pmc_grch38scale.csv", "r04_pmc_grch38scale.csv", "r03_pmc_grch38scale.csv")}
                                     .get(float(args.genome_mbp), ())) if os.path.exists(os.path.join(ROOT, "profiles", n))), None)
        tab = pmc_table(pmc_name) if (pmc_name and default_run) else None
        if tab:      # do the stored counters describe THESE kernels?  (tools/kernel_hash.py beside the table; none stored = unknown = stale)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from kernel_hash import kernel_sources_hash
            try:
                traffic_stale = open(os.path.join(ROOT, "profiles", pmc_name + ".srchash")).read().split()[0] != kernel_sources_hash()
            except (OSError, IndexError):
                traffic_stale = True
        k1_name = next((k for k in ("ema_k_seed_t<false;false>", "ema_k_seed_t<false>", "ema_k_seed") if tab and (k, "FETCH_SIZE") in tab and (k, "WRITE_SIZE") in tab), None)
        if k1_name:      # (the product build of the template -- [r5] without pass 3 -- or the plain kernel of older profiles)
            series = eng.seed_launches_per_series()
            kb = sum(tab[(k1_name, c)][0] / (tab[(k1_name, c)][1] / series) for c in ("FETCH_SIZE", "WRITE_SIZE"))
            if k1_name == "ema_k_seed_t<false;false>" and ("ema_k_seed_p3", "FETCH_SIZE") in tab:      # [r5] pass 3's own kernel, one launch behind every series: part of the same interval of HIP events
                kb += sum(tab[("ema_k_seed_p3", c)][0] / tab[("ema_k_seed_p3", c)][1] for c in ("FETCH_SIZE", "WRITE_SIZE") if ("ema_k_seed_p3", c) in tab)
            traffic = int(kb * 1024)
            traffic_source = f"profiles/{pmc_name} (separate FETCH_SIZE and WRITE_SIZE passes of this command; stored, not measured in this run)"
        roofline = {"bound": "hbm", "kernel": "ema_k_seed", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source, "traffic_source_stale": traffic_stale,
                    "algorithmic_bytes_per_launch": int(k1_bytes), "kernel_ms": round(dom_ms, 3),
                    "note": f"one launch (series) = one of {n_slices} slices of a batch; mean over the {args.steps} timed steps, in which "
                            f"launches of different slices, kernels and batches run concurrently and share the chip -- 'isolated' is the "
                            f"same launch with the chip to itself.  K2..K4 algorithmic bytes per launch: {int(rest_bytes)}",
                    "all_kernels_ms": {k: round(v, 3) for k, v in kernel_ms.items()}}
        if kernel_ms_isolated:
            iso_ms = kernel_ms_isolated["seed_ms"]
            isolated = k1_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
            roofline["isolated"] = {"achieved": round(isolated, 2), "frac": round(isolated / HBM_PEAK_GBS, 5), "kernel_ms": round(iso_ms, 3)}
            roofline["all_kernels_ms_isolated"] = {k: round(v, 3) for k, v in kernel_ms_isolated.items()}
        # K2b: bound by instruction issue, not by HBM -- wave-level VALU instructions per second against the chip's issue peak
        roofline_k2b = None
        # (K2b, K2c and K2d are builds of one template, ema_k_align_t<SMALL;AVL;WPS;MODE>: their instructions are summed per K2b launch)
        k2_names = sorted({k for k, c in (tab or {}) if c == "SQ_INSTS_VALU" and (k == "ema_k_align" or k.startswith("ema_k_align_t<"))})
        def k2_mode(k):      # ema_k_align_t<SMALL;AVL;WPS;MODE;PROF>
            a = k[k.index("<") + 1:k.rindex(">")].split(";") if "<" in k else []
            return a[3] if len(a) > 3 else "0"
        k2_main = [k for k in k2_names if k2_mode(k) == "0"]
        if k2_names and k2_main and kernel_ms_isolated and not traffic_stale:      # (a stale instruction count would describe other kernels)
            insts = sum(tab[(k, "SQ_INSTS_VALU")][0] for k in k2_names)
            launches = max(tab[(k, "SQ_INSTS_VALU")][1] for k in k2_main)
            per_launch = insts / launches
            ms = kernel_ms_isolated["extend_ms"]
            ach = per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            roofline_k2b = {"bound": "valu-issue", "kernel": " + ".join(k2_names) + " (+ ema_k_align_simple in kernel_ms)", "achieved": round(ach, 2),
                            "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave-instructions/s", "frac": round(ach / VALU_PEAK_GINST, 5),
                            "insts_per_launch": int(per_launch), "kernel_ms": round(ms, 3),
                            "source": f"profiles/{pmc_name}: SQ_INSTS_VALU per launch (stored PMC pass) / this run's isolated K2 time; peak = "
                                      f"the measured chip-wide rate of full-rate 32-bit integer VALU instructions at 8 waves per SIMD (tools/valu_issue_microbench.hip, profiles/r06_valu_issue_microbench.txt; half-rate instructions -- v_bcnt, 64-bit shifts -- peak at 593 G)"}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (contract)
            import resource
            cores, model = one_socket_cpus()
            # How many CPUs does this job really get?  The GPU boxes show a 64-core socket but schedule the job on a CPU quota
            # (r02b: throughput flat from 16 threads on, CPU seconds / wall seconds = 16.0 at 32 and 64 threads): a short probe with
            # one thread per core of the socket measures it, and the baseline then runs with that many threads on that many cores.
            probe = batches[0].subset(0, min(batches[0].n, 300 * len(cores)))
            with pinned_to(cores):
                r0, w0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
                O.bench_pairs(idx, opt, probe.bases, probe.off, len(cores))
                r1, w1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
            granted = (r1.ru_utime + r1.ru_stime - r0.ru_utime - r0.ru_stime) / max(1e-9, w1 - w0)
            n_thr = max(1, min(len(cores), int(granted + 0.5)))
            use = cores[:n_thr]
            rate_1 = n_1 / t_1
            n_s = min(args.cpu_sample, batches[0].n, int(25 * n_thr * rate_1))      # about 25 s of work
            sample = batches[0].subset(0, n_s)
            with pinned_to(use):
                secs, _ = O.bench_pairs(idx, opt, sample.bases, sample.off, n_thr)
            cpu = {"value": round(n_s / secs, 1), "unit": "pairs/s", "cores": n_thr, "kind": "port",
                   "one_thread_pairs_per_s": round(rate_1, 1), "parallel_efficiency": round((n_s / secs) / (n_thr * rate_1), 3),
                   "cpu": model, "socket_cores": len(cores), "cpus_granted_to_the_job": round(granted, 1),
                   "sample": f"first {n_s} pairs of batch 0, oracle/ (CPU restatement, not upstream bwa), {n_thr} OpenMP threads on {n_thr} physical "
                             f"cores of NUMA node 0 (the socket has {len(cores)}; a probe with one thread per core got {granted:.1f} CPUs' worth of "
                             f"run time from the host), {secs:.1f}s; 1 thread on {n_1} pairs: {rate_1:.1f} pairs/s"}
        total_pairs = int(gathered[:, 0].sum())
        value = total_pairs / elapsed
        out = {
            "metric": "aligned read-pairs/sec (2x150 bp) on the seed-and-extend hot path", "value": round(value, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32/u64 (integer DP, FM-index ranks)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: 10x-style FR pairs, 2x150 bp sequenced = R1 127 bp after the 16 bp barcode + 7 bp trim "
                                   f"(reference cpp/correct.cc:550) and R2 150 bp, 0.5% subs, 0.05% indels, 1% chimeric; {args.steps} steps over "
                                   f"{n_batches} distinct batches of {args.pairs} pairs per GPU (one barcode bucket per GPU), inputs staged in HBM "
                                   f"before the timed region, candidates + append_alignments records delivered to host memory inside it "
                                   f"(passes queued up to three deep, result layout and packing on the device); "
                                   f"{fallback}reference = {gname} with injected repeat families (default: GRCh38-scale, 3.1 Gbp; genome by "
                                   f"csrc/synth_genome.cpp, index by ema_index_build with the suffix array sorted on the GPU, both before the timed "
                                   f"region; --genome-mbp 0 = chr20-scale)",
                       "pairs_per_step_per_gpu": args.pairs, "distinct_batches": n_batches, "buffer_sets": n_sets, "max_occ": 3000,
                       "parallelism": f"buckets x{world}", "engine_knobs_in_env": knobs,
                       "method": {"warmup_passes": args.warmup, "since": "r04: three warm-up passes by default (r01-r03: one), so that the three pooled "
                                  "page-locked batch sets exist before the clock starts; host CPU figures leave out the bench's own spot-check "
                                  "sampling (host.spot_check_sampling_cpu_s_excluded); the spot check covers rank 0's kept samples -- r04+ figures "
                                  "are not directly comparable with BENCH_r01-r03",
                                  "cold_first_pass_s": None if cold_first_pass_s is None else round(cold_first_pass_s, 3),
                                  "ms_between_steps_reaching_the_sink": step_gaps}},
            "boundary": boundary, "engine_resident": resident,
            "roofline": roofline, "roofline_k2b": roofline_k2b, "cpu_baseline": cpu,
            "bucket_stats": {f: int(gathered[:, i].sum()) for i, f in enumerate(shard.STAT_FIELDS)},
        }
        out["bucket_stats"].update(capacity_flags=int(any_flag), oracle_spot_check_pairs=int(n_checked), oracle_spot_check_pairs_of_timed_steps=int(n_checked_timed), oracle_spot_check_full_tier_pairs=int(n_checked_full),
                                   oracle_spot_check_mismatches=int(bad))
        # What the host costs, and what that predicts for N ranks on this node's CPU grant: every rank needs its own host threads for
        # fetch assembly and the append stage, and the node grants the job a fixed number of CPUs whatever N is.
        cpu_s_per_pair = host_cpu_s / float(args.pairs * args.steps)
        per_gpu = value / world
        out["host"] = {"cpu_seconds_per_million_pairs": round(cpu_s_per_pair * 1e6, 3), "what": "CPU seconds (user + system) of one rank's process over the "
                       "timed region: download of the device-made batch layout and the append_alignments stage on the host's threads (the bench's own "
                       "spot-check sampling in the sink is measured and left out)",
                       "cpus_granted_to_the_node": node_cpus, "host_threads_per_rank": int(os.environ.get("EMA_HOST_THREADS", "0")) or min(32, node_cpus),
                       "engine_open_s": round(open_s, 2), **stock, "spot_check_sampling_cpu_s_excluded": round(sink_cpu[0], 3),
                       "d2h_candidate_bytes_per_pair": round(112 * out["bucket_stats"]["candidates"] / max(1, total_pairs) + 2 * (8 + 8 + 4), 1)}      # 112-byte candidates + per-read layout; CIGAR operations (4 bytes each) come on top
        out["scaling_prediction"] = {
            "status": "PREDICTED from this run's per-GPU rate and host CPU cost; no multi-GPU run has been measured on hardware",
            "per_n_gpus": {str(n): {"gpu_bound_pairs_per_s": round(n * per_gpu, 1),
                                    "host_bound_pairs_per_s": round(node_cpus / cpu_s_per_pair, 1) if cpu_s_per_pair > 0 else None,
                                    "predicted_pairs_per_s": round(min(n * per_gpu, node_cpus / cpu_s_per_pair if cpu_s_per_pair > 0 else n * per_gpu), 1)}
                           for n in (1, 2, 4, 8)},
            "note": "host_bound divides the CPUs granted to the node by ONE rank's CPU seconds per pair over its timed region (all of the process's "
                    "threads, the Python sink included): a rough bound"}
        out["bucket_files_to_sam"] = sam_leg(args, eng, batches, workdir, world) if not (args.no_extras or args.no_sam_leg) else None
        if out["bucket_files_to_sam"]:
            # The same prediction for the end-to-end leg: N ranks run N such streams, every stream's host stages
            # (reader, staging, fetch, append, clouds / EM / duplicates, formatter) drawing on the ONE CPU grant of the node.
            leg = out["bucket_files_to_sam"]
            cpu_per_pair = leg["host_cpu_seconds_per_million_pairs"] / 1e6
            rate1 = leg["value"]
            leg["scaling_prediction"] = {
                "status": "PREDICTED from this run's 1-GPU rate and host CPU cost; no multi-GPU run has been measured on hardware",
                "per_n_gpus": {str(n): {"gpu_bound_pairs_per_s": round(n * rate1, 1),
                                        "host_bound_pairs_per_s": round(node_cpus / cpu_per_pair, 1) if cpu_per_pair > 0 else None,
                                        "predicted_pairs_per_s": round(min(n * rate1, node_cpus / cpu_per_pair if cpu_per_pair > 0 else n * rate1), 1)}
                               for n in (1, 2, 4, 8)},
                "cpus_granted_to_the_node": node_cpus,
                "cpus_for_6x_at_8_gpus": int(np.ceil(6.0 * rate1 * cpu_per_pair)) if cpu_per_pair > 0 else None,
                "note": "north_star asks for >= 6x the 1-GPU rate at 8 GPUs: that takes cpus_for_6x_at_8_gpus CPUs' worth of host threads at this "
                        "run's CPU cost per pair; on cpus_granted_to_the_node the stream is host-bound from the N where gpu_bound exceeds host_bound"}
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
