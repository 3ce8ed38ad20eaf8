"""bench.py's distributed control flow on hardware.  A one-GPU box cannot run N > 1, but it can run everything
the N > 1 path is made of: the launcher the driver uses (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`),
the `nccl` (RCCL) process group bound to cuda:0, the barriers around the timed region, agree()'s all-reduces on device tensors for
the maximum over ranks and the fatal exits, and shard.gather_stats(device="cuda") after a timed region -- with `--dist-at-world-1`,
which makes bench.py take those branches at world size 1.  The launcher is a CHILD of this test, started before this process has
touched the GPU for it (the pool forbids replacing a GPU-initialised process).  No scaling curve is measured by this: README."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_distributed_control_flow_under_the_launcher_at_world_size_1(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", EMA_BENCH_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "EMA_BENCH_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--genome-mbp", "0", "--pairs", "16384", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-sam-leg", "--dist-at-world-1"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "pairs/s"
    assert d["value"] > 0 and abs(d["value"] - 16384 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-3 * d["value"]
    st = d["bucket_stats"]
    assert st["capacity_flags"] == 0 and st["oracle_spot_check_mismatches"] == 0 and st["oracle_spot_check_pairs"] > 0
    assert st["pairs"] == 16384 * 2      # the all-gathered table (one bucket per step and rank), summed
    assert "process group" in p.stderr or "nccl" in p.stderr.lower()      # bench.py logs the backend it initialised
