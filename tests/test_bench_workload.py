"""bench.py's workload step with more than one rank, on CPU (gloo, 127.0.0.1): rank 0 builds genome and index while the
others wait at the barrier; a second run on the same directory reuses both; every rank simulates its own batches from
the one genome.npy (memory-mapped by the worker processes: no rank regenerates the genome).  (The timed part of bench.py needs a GPU; this is the part the driver's N = 2, 4, 8 runs share a
directory for.)"""
import os
import socket
import types

import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, workdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    args = types.SimpleNamespace(genome_mbp=1.2, pairs=300)
    out = []
    for _ in range(2):      # first pass builds (rank 0 only), second pass finds everything in place
        if rank == 0:
            bench.build_reference(args, workdir)
        dist.barrier()
        batches = bench.make_batches(args, rank, world, workdir, 2)
        out.append(([p.n for p in batches], int(batches[0].off[-1]), batches[0].read(0), batches[1].read(0), sorted(os.listdir(workdir))))
        dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_ranks_share_one_workload_directory(tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        first, second = results[rank]
        assert first[:4] == second[:4]                       # the cached genome and batches give the same reads again
        assert first[0] == [300, 300] and first[2] != first[3]      # two distinct batches per rank
        for ext in (".bwt", ".sa", ".fsa", ".pac", ".ann", ".amb", ".stamp", ".gstamp"):
            assert "ref.fa" + ext in second[4]
        assert "genome.npy" in second[4]
        assert sum(f.startswith("reads_") for f in second[4]) == 4      # 2 ranks x 2 batches, cached
    assert results[0][0][2] != results[1][0][2]              # different ranks, different buckets
