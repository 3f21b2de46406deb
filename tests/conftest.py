import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box); parity only, no assertion on elapsed time")
    config.addinivalue_line("markers", "gpuperf: needs a real MI355X; asserts wall-clock figures (latency SLO, keep-up, concurrency) -- "
                                       "run with -m gpuperf on a quiet box, never part of -m gpu")


def pytest_collection_modifyitems(config, items):
    """`gpuperf` tests (wall-clock assertions) run ONLY when the -m expression names gpuperf: `-m gpu` -- the parity gate the
    driver runs -- must not turn red because a box was noisy, and `-m "not gpu"` has no GPU at all."""
    if "gpuperf" in (config.option.markexpr or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("gpuperf") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def refdata(golden_dir):
    return os.path.join(golden_dir, "reference_data")
