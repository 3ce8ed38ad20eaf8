import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def tuning():
    """The library's development knobs for one test (ema_amd.engine.set_tuning -> ema_engine_set_tuning): `tuning(seed_tail=0)`;
    cleared again when the test ends.  Replaces one environment variable per knob."""
    from ema_amd import engine
    engine.set_tuning()
    yield engine.set_tuning
    engine.set_tuning()
