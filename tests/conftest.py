import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests fail loudly (never skip silently) when selected on a box without a GPU or the extension."""
    return


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
        return cache[name]
    return load
