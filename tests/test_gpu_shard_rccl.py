"""The RCCL branch of the statistics gather on hardware: a child process -- started before this process
touches the GPU for it, the pool forbids replacing a GPU-initialised process -- initialises torch.distributed with the `nccl`
backend (RCCL on ROCm) at world size 1 on cuda:0 and runs ema_amd.shard.gather_stats(..., device="cuda"): the dist.all_gather
of ema_amd/shard.py executes on an MI355X.  World size 1 is what a one-GPU box can offer; no scaling curve exists (README)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json, datetime
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["EMA_ROOT"])
from ema_amd import shard
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(minutes=5))
n = 5
local = np.array([[1000 * (f + 1) + b for f in range(len(shard.STAT_FIELDS))] for b in range(n)], dtype=np.int64)
table = shard.gather_stats(local, n, device="cuda")
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
print(json.dumps({"backend": dist.get_backend(), "table": table.tolist(), "allreduce": t.tolist()}))
dist.barrier()
dist.destroy_process_group()
'''


def test_rccl_all_gather_of_bucket_statistics_at_world_size_1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "EMA_ROOT": ROOT})
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["backend"] == "nccl"
    from ema_amd import shard
    assert d["table"] == [[1000 * (f + 1) + b for f in range(len(shard.STAT_FIELDS))] for b in range(5)]      # every field of the record, bucket by bucket
    assert d["allreduce"] == [1.0] * 4
