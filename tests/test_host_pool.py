"""The host stages' one thread pool (ema_amd/csrc/host_pool.h) and their CPU accounting (host_cpuacct.h, ema_host_cpu_seconds):
passes started from inside another pass's piece and from several caller threads at once complete with the right results under
ThreadSanitizer; the accounting charges the bucket reader's CPU time to the reader and can be reset."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_pool_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "pool_test")
    src = os.path.join(ROOT, "tests", "native", "pool_test.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-I" + os.path.join(ROOT, "ema_amd", "csrc"), "-o", exe, src, "-lpthread"])
    env = dict(os.environ, EMA_HOST_THREADS="6", TSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([exe], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "ThreadSanitizer" not in out.stderr
    total, last, threads = out.stdout.split()[:3]
    assert int(threads) == 6 and int(last) == 390000      # sum over 100 rounds of sum(i * 5 for i < 40)
    # nested passes: every (i, j) piece ran exactly once
    want = 200 * sum(sum((i * j + k) % 7 for k in range(1000)) for i in range(64) for j in range(8))
    assert int(total) == want


def test_host_cpu_seconds_charges_the_reader(tmp_path):
    from ema_amd import ingest, stream, synth
    from common import small_ref
    _, ctg = small_ref("two_contigs")
    pairs = synth.make_pairs(ctg, 20000, seed=3)
    path = str(tmp_path / "bucket")
    synth.write_special_fastq_fixed(path, pairs)
    stream.host_cpu_seconds(reset=True)
    b = ingest.read_bucket(path)
    assert b.n_pairs == 20000
    cpu = stream.host_cpu_seconds()
    assert cpu["reader"] > 0 and all(v == 0 for k, v in cpu.items() if k != "reader"), cpu
    assert stream.host_cpu_seconds(reset=True)["reader"] > 0 and stream.host_cpu_seconds()["reader"] == 0
