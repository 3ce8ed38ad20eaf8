#!/usr/bin/env python3
"""bench.py -- aligned read pairs/s of the `ema align` seed-and-extend hot path on MI355X.

One "step" = one pass of the whole hot path (K1 seeding -> K2 chaining/extension -> K3 mate rescue -> K4 final
alignment) over one batch of synthetic read pairs that is already resident in HBM, i.e. exactly the work the
reference does per pair in append_alignments() (reference src/align.c:1005-1038: bwa_mem_mate_sw + one
bwa_smith_waterman per candidate), for `pairs_per_step` pairs.  Index build, read generation, upload and result
download are outside the timed region.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

Multi-GPU: barcode buckets are independent, so every rank aligns its own bucket of the same size
against its own replica of the index -- weak scaling, no collective on the data path; the per-bucket statistics
(pairs, candidates, mapped mates) are gathered over RCCL at the end.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline     : HBM roofline of the dominant kernel (algorithmic bytes from the oracle's instrumentation of the
                 same reads / that kernel's mean launch time from HIP events on the engine's stream)
  cpu_baseline : the oracle (a CPU restatement, NOT upstream bwa: parity unpinned) timed on the host cores on a
                 bounded sample of the same workload.
"""
import argparse
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before anything initialises the GPU runtime: see ema_amd/csrc/engine.hip
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured streaming copy)
CHR20_LEN = 64_444_167


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(args, rank, world, workdir):
    """Synthetic genome + index (built once, by rank 0) and this rank's bucket of read pairs."""
    from ema_amd import synth, build_index
    prefix = os.path.join(workdir, "ref.fa")
    if args.genome_mbp > 0:
        n_ctg = max(1, int(np.ceil(args.genome_mbp / 155.0)))          # chromosome-sized contigs (contig lengths are 32-bit)
        lens = [int(args.genome_mbp * 1e6 / n_ctg)] * n_ctg
        gname = f"synthetic {args.genome_mbp:g} Mbp, {n_ctg} contig{'s' if n_ctg > 1 else ''}"
    else:
        lens = [CHR20_LEN]
        gname = "synthetic chr20-scale (64,444,167 bp, 1 contig)"
    t = time.time()
    stamp = json.dumps({"lens": lens, "seed": synth.GENOME_SEED, "builder": 2})
    gpath = os.path.join(workdir, "genome.npy")
    try:      # the genome of an earlier run on this box, if it is the same one
        have_genome = open(prefix + ".gstamp").read() == stamp and os.path.exists(gpath)
    except OSError:
        have_genome = False
    if have_genome:
        flat = np.load(gpath, mmap_mode="r")
        ctg, at = [], 0
        for n in lens:
            ctg.append(np.asarray(flat[at:at + n])); at += n
        log(f"[rank {rank}] genome {gname}: loaded in {time.time() - t:.1f}s")
    else:
        ctg = synth.make_genome(lens, seed=synth.GENOME_SEED)
        log(f"[rank {rank}] genome {gname}: {time.time() - t:.1f}s")
    if rank == 0:
        # the index of an earlier run on this box (same genome, same builder) is reused: the driver's N = 1, 2, 4, 8 runs
        # come back to back on one node
        if not have_genome:
            if os.path.exists(prefix + ".gstamp"):
                os.remove(prefix + ".gstamp")
            np.save(gpath, np.concatenate(ctg))
            with open(prefix + ".gstamp", "w") as f:
                f.write(stamp)
        have = all(os.path.exists(prefix + e) for e in (".bwt", ".sa", ".fsa", ".pac", ".ann", ".amb", ".stamp"))
        if have and open(prefix + ".stamp").read() == stamp:
            log("[rank 0] index of this genome found in " + workdir + ": reused")
        else:
            t = time.time()
            if os.path.exists(prefix + ".stamp"):
                os.remove(prefix + ".stamp")
            synth.write_fasta(prefix, ctg, names=["chr20"] if len(ctg) == 1 else [f"chr{i + 1}" for i in range(len(ctg))])
            build_index(prefix)
            with open(prefix + ".stamp", "w") as f:
                f.write(stamp)
            log(f"[rank 0] index built in {time.time() - t:.1f}s")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t = time.time()
    pairs = synth.make_pairs(ctg, args.pairs, seed=synth.READS_SEED + 7919 * rank, len1=127, len2=150)
    log(f"[rank {rank}] {args.pairs} pairs simulated in {time.time() - t:.1f}s")
    return prefix, pairs, gname


def algorithmic_bytes(stats, sa_width):
    """bytes the algorithm must move per unit, from the oracle's counters on the same reads.
    K1 (seeding): two 32-byte rank blocks per bwt_extend + the read.  K2..K4: SA rows, reference windows (2 bit/base),
    region records, CIGAR ops."""
    k1 = 64 * stats["n_ext"] + stats["l_read"]
    rest = sa_width * stats["n_occ"] + stats["w_ref"] // 4 + 88 * stats["n_regs"] + 4 * stats["n_cigar"]
    return k1, rest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="default: 10 steps x 1 Mi pairs = the 10 M pairs of BASELINE configs[1]")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1048576, help="pairs per step and per GPU (one resident batch)")
    ap.add_argument("--sync-each-step", action="store_true", help="wait for every step before queueing the next (no overlap of step tails)")
    ap.add_argument("--genome-mbp", type=float, default=3100.0,
                    help="size of the synthetic reference; default GRCh38-scale (3.1 Gbp in 20 contigs, BASELINE configs[1]); "
                         "0 = chr20-scale (64.4 Mbp, one contig: configs[0]'s reference)")
    ap.add_argument("--cpu-sample", type=int, default=400000, help="pairs of the same workload timed on the host CPU")
    ap.add_argument("--streams", type=int, default=0, help="slices of a batch on their own HIP streams (0 = engine default)")
    ap.add_argument("--lean-seed-extends", type=int, default=0, help="engine option lean_seed_extends (0 = engine default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-capacity-flags", action="store_true", help="print the (invalid) line even if reads overflowed an engine capacity")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        import datetime
        # rank 0 builds genome and index (minutes at the default scale) while the others wait at a barrier
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=30))
    assert world == max(1, args.gpus) or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    import __graft_entry__
    if rank == 0:
        __graft_entry__.ensure_built()
    if world > 1:
        dist.barrier()

    import tempfile
    workdir = os.environ.get("EMA_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "ema_bench_%d" % os.getuid())
    os.makedirs(workdir, exist_ok=True)
    # A reference of G bases needs ~21 G bytes of files (flat suffix array: 16 G) and ~26 G bytes of host memory while the
    # index is built.  A box that cannot hold the requested reference gets the chr20-scale one instead, and the JSON says so
    # -- better a line on the smaller workload, labelled as such, than none.
    fallback = ""
    if rank == 0 and args.genome_mbp > 0 and not os.path.exists(os.path.join(workdir, "ref.fa.stamp")):
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
    if world > 1:      # rank 0 decides for everybody (a plain tensor collective, like the ones below)
        g = torch.tensor([args.genome_mbp if rank == 0 else 1e18], dtype=torch.float64, device="cuda")
        dist.all_reduce(g, op=dist.ReduceOp.MIN)
        args.genome_mbp = float(g.item())
    prefix, pairs, gname = build_workload(args, rank, world, workdir)

    from ema_amd.engine import Engine, default_opts
    o = default_opts()
    o.batch_pairs = args.pairs
    o.n_streams = args.streams
    o.lean_seed_extends = args.lean_seed_extends
    t = time.time()
    eng = Engine(prefix, device=local, opts=o)
    log(f"[rank {rank}] engine open (index in HBM) {time.time() - t:.1f}s")
    eng.stage(pairs.bases, pairs.off)          # nt4 conversion + H2D: outside the timed region

    def sync_all():
        eng.sync()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        eng.run()
    sync_all()
    # Timed region: K steps queued back to back (as a host streaming buckets would), one wait at the end; every step is
    # the complete K1..K4 pass, both capacity tiers included.
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run()
        if args.sync_each_step:
            eng.sync()
    sync_all()
    elapsed = time.perf_counter() - t0
    # per-launch kernel durations (HIP events on the launching streams): the last step of the timed region, where launches
    # of different slices overlap, and one extra untimed pass with the slices one after another (isolated launches)
    tm = eng.timing()
    kernel_ms = {k: tm[k] for k in ("seed_ms", "extend_ms", "rescue_ms", "final_ms", "total_ms", "full_tier_ms")}
    log(f"[rank {rank}] full-capacity tier K1..K4 ms: {tm['full_ms']}")
    eng.run(serial=True)
    eng.sync()
    tm = eng.timing()
    kernel_ms_isolated = {k: tm[k] for k in kernel_ms}
    n_slices = eng.n_streams
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # results of the last step: parity spot check against the oracle + bucket statistics
    batch = eng.fetch(allow_limit=True)
    if batch.status.max() != 0:
        flags, counts = np.unique(batch.status[batch.status != 0], return_counts=True)
        log(f"[rank {rank}] ERROR: reads exceeded an engine capacity (status flag: count) {dict(zip(flags.tolist(), counts.tolist()))}: "
            f"their pairs have no candidates, so the timed steps skipped work and the number is not valid")
    any_flag = int(batch.status.max() != 0) if len(batch.status) else 0
    if world > 1:      # every rank must take the same exit
        tf = torch.tensor([any_flag], dtype=torch.int32, device="cuda")
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        any_flag = int(tf.item())
    if any_flag and not args.allow_capacity_flags:
        eng.close()
        if world > 1:
            dist.destroy_process_group()
        raise SystemExit(2)
    from ema_amd import shard
    stats_vec = shard.bucket_stats(batch, pairs.n)
    # the "trivial RCCL gather of per-bucket statistics" of the north star: one bucket per rank, O(100 B) over xGMI
    gathered = shard.gather_stats(stats_vec[None, :], world, device=("cuda" if world > 1 else None))

    out = None
    if rank == 0:
        import oracle_lib as O
        idx, opt = O.Index(prefix), O.default_opt()
        # spot check: the first pairs of the bucket must equal the oracle bit for bit
        n_chk = min(200, pairs.n)
        bad = 0
        for p in range(n_chk):
            ref = O.align_pair(idx, opt, pairs.read(2 * p), pairs.read(2 * p + 1))
            for m in range(2):
                got = [(int(c["rb"]), int(c["re"]), int(c["qb"]), int(c["qe"]), int(c["score"]), int(c["pos"]), int(c["NM"]),
                        batch.cigar_of(c).tolist()) for c in batch.mate(p, m)]
                exp = [(d["rb"], d["re"], d["qb"], d["qe"], d["score"], d["pos"], d["NM"], d["cigar"]) for d in ref[m]]
                bad += got != exp
        if bad:
            log(f"WARNING: {bad} reads of the {n_chk}-pair spot check differ from the oracle")
        # algorithmic bytes: the oracle's counters on a sample of the SAME reads, scaled to the batch
        n_s = min(args.cpu_sample, pairs.n)
        sample = pairs.subset(0, n_s)
        O.stats_reset()
        t_1, _ = O.bench_pairs(idx, opt, sample.bases[:sample.off[2 * min(n_s, 4000)]], sample.off[:2 * min(n_s, 4000) + 1], 1)
        st = O.stats_get()
        scale = pairs.n / float(min(n_s, 4000))
        k1_bytes, rest_bytes = algorithmic_bytes(st, 4)
        k1_bytes *= scale / n_slices           # one launch covers one slice of the batch
        rest_bytes *= scale / n_slices
        # The HBM roofline is meaningful for K1 only: it is the kernel whose work is FM-index gathers (K2..K4 move ~100x
        # fewer algorithmic bytes and are bound by instruction issue; see DESIGN.md and profiles/), so it is the kernel
        # reported here whichever launch is longer.
        dom_bytes = k1_bytes
        dom_ms, iso_ms = kernel_ms["seed_ms"], kernel_ms_isolated["seed_ms"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        isolated = dom_bytes / (iso_ms * 1e-3) / 1e9 if iso_ms > 0 else 0.0
        # HBM traffic of the same kernel: not measurable from inside this process -- taken from the committed PMC passes of this
        # very command (profiles/, separate FETCH_SIZE and WRITE_SIZE runs; KB per launch, three launches per series), and
        # only when this run is that default workload.
        traffic = None
        pmc_name = {0.0: "r01f_pmc_chr20_1Mpairs.csv", 3100.0: "r01g_pmc_grch38scale_1Mpairs.csv"}.get(float(args.genome_mbp), "none")
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_name)
        if args.pairs == 1048576 and n_slices == 3 and args.lean_seed_extends == 0 and os.path.exists(pmc):
            import csv
            kb = {r["counter"]: float(r["sum_over_run"]) / (float(r["launches"]) / 3.0) for r in csv.DictReader(open(pmc))
                  if r["kernel"] == "ema_k_seed" and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE")}
            if len(kb) == 2:
                traffic = int((kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024)
        roofline = {"bound": "hbm", "kernel": "ema_k_seed",
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(dom_bytes), "kernel_ms": round(dom_ms, 3),
                    "note": f"one launch = one of {n_slices} slices of the batch; in the timed region launches of different slices "
                            f"and kernels run concurrently and share the chip, so the per-launch rate understates the kernel: "
                            f"'isolated' is the same launch with the chip to itself.  "
                            + ("On this reference the rank structure (64 MB) is cache-resident and the kernel is bound by "
                               "instruction issue, not by HBM (profiles/: FETCH_SIZE per launch ~ algorithmic bytes; ~60% of SIMD "
                               "cycles issue VALU work).  " if args.genome_mbp == 0 else
                               "On this reference the rank structure (3.1 GB) and the suffix array (50 GB) are far beyond the 256 MB "
                               "Infinity Cache: every rank query is a dependent 32-byte gather from HBM, so the kernel is bound by "
                               "gather latency x the waves in flight, not by bandwidth.  By time the largest kernel here is K2 "
                               "(extend_ms: chaining + banded extension, instruction-bound, ~0.2 GB of algorithmic bytes); the "
                               "roofline object stays on K1, the kernel that moves the bytes.  ")
                            + f"K2..K4 algorithmic bytes per launch: {int(rest_bytes)}",
                    "isolated": {"achieved": round(isolated, 2), "frac": round(isolated / HBM_PEAK_GBS, 5), "kernel_ms": round(iso_ms, 3)},
                    "all_kernels_ms": {k: round(v, 3) for k, v in kernel_ms.items()},
                    "all_kernels_ms_isolated": {k: round(v, 3) for k, v in kernel_ms_isolated.items()}}
        # which stage takes longest when a slice has the chip to itself (the roofline object above stays on K1, the stage that
        # moves the bytes; the others are bound by instruction issue and dependent latencies)
        stage_ms = {"K1 seeding": kernel_ms_isolated.get("seed_ms", 0.0), "K2 chaining + extension": kernel_ms_isolated.get("extend_ms", 0.0),
                    "K3 mate rescue": kernel_ms_isolated.get("rescue_ms", 0.0), "K4 final alignment": kernel_ms_isolated.get("final_ms", 0.0)}
        top = max(stage_ms, key=stage_ms.get)
        roofline["largest_stage_isolated"] = {"stage": top, "ms": round(stage_ms[top], 3)}
        cpu = None
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (contract)
            cores = len(os.sched_getaffinity(0))
            secs, _ = O.bench_pairs(idx, opt, sample.bases, sample.off, cores)
            cpu = {"value": round(n_s / secs, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
                   "sample": f"first {n_s} pairs of the same bucket, oracle/ (CPU restatement, not upstream bwa), "
                             f"{cores} OpenMP threads, {secs:.1f}s; 1 thread: {min(n_s, 4000) / t_1:.1f} pairs/s"}
        total_pairs = int(gathered[:, 0].sum()) * args.steps
        value = total_pairs / elapsed
        out = {
            "metric": "aligned read-pairs/sec (2x150 bp) on the seed-and-extend hot path", "value": round(value, 1),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32/u64 (integer DP, FM-index ranks)", "data": "synthetic",
            "config": {"workload": f"10x-style FR pairs R1=127 bp (150-16-7) R2=150 bp, 0.5% subs, 0.05% indels, 1% chimeric; "
                                   f"{args.pairs} pairs per GPU per step, one barcode bucket per GPU, resident in HBM; "
                                   f"{fallback}reference = {gname} with injected repeat families "
                                   f"(default: GRCh38-scale, 3.1 Gbp, as BASELINE configs[1] names; index built on the host cores in "
                                   f"~2 min before the timed region; --genome-mbp 0 = chr20-scale)",
                       "pairs_per_step_per_gpu": args.pairs, "max_occ": 3000, "parallelism": f"buckets x{world}"},
            "roofline": roofline, "cpu_baseline": cpu,
            "bucket_stats": {"pairs": int(gathered[:, 0].sum()), "candidates": int(gathered[:, 1].sum()),
                             "reads_with_candidates": int(gathered[:, 2].sum()), "capacity_flags": int(gathered[:, 3].max()),
                             "oracle_spot_check_mismatches": int(bad), "full_tier_pairs_rank0": int(batch.n_redone)},
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
