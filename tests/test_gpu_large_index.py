"""The GRCh38-scale index paths on a small reference ("next round" 1a): several rank superblocks and
8-byte suffix-array rows.  Each case is a child pytest over tests/large_index_cases.py with the test build of the
engine and/or an index written with 8-byte rows; see that file."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra):
    if "EMA_ENGINE_LIB" in env_extra and not os.path.exists(os.path.join(ROOT, "ema_amd", env_extra["EMA_ENGINE_LIB"])):
        subprocess.check_call(["make", "-C", ROOT, "-j", str(min(8, os.cpu_count() or 1)), "test-libs"])      # normally travels with the snapshot
    env = dict(os.environ)
    env.update(env_extra)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "large_index_cases.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]


def test_several_rank_superblocks():
    _run({"EMA_ENGINE_LIB": "libema_engine_ss16.so"})


def test_eight_byte_suffix_array_rows():
    _run({"EMA_INDEX_SA64": "1"})


def test_both_together():
    _run({"EMA_ENGINE_LIB": "libema_engine_ss16.so", "EMA_INDEX_SA64": "1"})
