import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 20260928  # same seed as tests/golden/make_golden.py


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_rng(tag):
    return np.random.default_rng(SEED + tag)


def relu_normal(g, shape):
    return np.maximum(g.standard_normal(shape, dtype=np.float32), 0.0)


@pytest.fixture(scope="session")
def golden():
    return load_golden
