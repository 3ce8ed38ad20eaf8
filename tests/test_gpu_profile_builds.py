"""The diagnostic builds of the kernels give the product's results.  K2b exists three times over -- the product build, the
diagnostic build with phase clocks and the per-read log (tuning knob phase_profile=1 / 2) and the product build with seven clocks in scalar
registers (phase_profile=3, what DESIGN.md section 3 argues from) -- and a number measured on a build is only worth quoting if that build computes
the same thing: a slice of the pipeline parity suite (clean reads, rescue, a repeat family with chain-rich reads set aside) runs
against the oracle under each setting, in a child process (the library reads the variable when an engine opens)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("level", ["1", "3"])
def test_profile_builds_are_parity_clean(level):
    env = dict(os.environ, EMA_TUNING="phase_profile=" + level)      # (the child's tests set no knobs of their own)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_pipeline.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "clean or noisy or repeat_family or few_mismatches"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    assert " passed" in p.stdout
