"""bench.py's whole control flow with two ranks on CPU (gloo, 127.0.0.1) and a stand-in engine (tests/bench_stub: candidates
from the oracle): reference built by rank 0 only, batches per rank, staged slots over two buffer sets, timed resident stream,
boundary and engine-only passes, spot check, statistics all-gather, ONE JSON line from rank 0 -- and the exits every rank
must take together (a capacity flag on one rank, a spot-check mismatch on rank 0).  The nccl branch differs only in the
backend name and the tensor device."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(tmp_path, extra_env=None, extra_args=()):
    env = dict(os.environ)
    env.update({"EMA_BENCH_BACKEND": "gloo", "EMA_BENCH_DIR": str(tmp_path), "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + ROOT})
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "24", "--genome-mbp", "0.6",
           "--no-cpu-baseline", "--engine-module", "bench_stub"] + list(extra_args)
    return subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)


def test_two_ranks_run_the_whole_bench_flow(tmp_path):
    p = _launch(tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["unit"] == "pairs/s"
    assert d["bucket_stats"]["pairs"] == 2 * 2 * 24 and d["bucket_stats"]["oracle_spot_check_mismatches"] == 0
    assert d["boundary"]["value"] > 0 and d["engine_resident"]["value"] > 0 and d["cpu_baseline"] is None
    assert d["config"]["buffer_sets"] == 1 and d["roofline"]["kernel"] == "ema_k_seed" and d["roofline"]["traffic"] is None


def test_two_sets_schedule(tmp_path):
    p = _launch(tmp_path, extra_args=["--two-sets"])
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["buffer_sets"] == 2 and d["bucket_stats"]["pairs"] == 2 * 2 * 24


def test_a_capacity_flag_on_one_rank_ends_every_rank(tmp_path):
    p = _launch(tmp_path, {"BENCH_STUB_FLAG_RANK": "1"})
    assert p.returncode != 0 and "exceeded an engine capacity" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_a_spot_check_mismatch_is_fatal(tmp_path):
    p = _launch(tmp_path, {"BENCH_STUB_CORRUPT": "1"})
    assert p.returncode != 0 and "differ from the oracle" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_gpus_flag_without_a_launcher_spawns_the_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torch.distributed.run must not run as a silent one-GPU job."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update({"EMA_BENCH_BACKEND": "gloo", "EMA_BENCH_DIR": str(tmp_path), "PYTHONPATH": os.path.join(ROOT, "tests") + os.pathsep + ROOT})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--pairs", "16", "--genome-mbp", "0.6",
                        "--no-cpu-baseline", "--no-extras", "--engine-module", "bench_stub"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and "launching" in p.stderr


def test_host_threads_are_partitioned_over_the_ranks_of_a_node():
    """N ranks share the CPUs the node grants the job: never more host threads in total than CPUs (round 2's floor of 4 per rank
    put 32 threads on 19 CPUs at 8 ranks), at least one each."""
    import bench
    for cpus in (8, 16, 19, 64, 192):
        for world in (1, 2, 4, 8):
            t = bench.host_threads_for_rank(cpus, world)
            assert 1 <= t <= 32
            assert t * world <= max(cpus, world)
    assert bench.host_threads_for_rank(19, 8) == 2 and bench.host_threads_for_rank(19, 1) == 19 and bench.host_threads_for_rank(256, 2) == 32
    assert 1 <= bench.cpus_granted() <= (os.cpu_count() or 1)
