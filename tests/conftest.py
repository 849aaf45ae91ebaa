import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.fixture(scope="session")
def golden_ops():
    return load_golden("ops.npz")


@pytest.fixture(scope="session")
def golden_modules():
    return load_golden("modules.npz")


@pytest.fixture(scope="session")
def golden_lbs():
    return load_golden("lbs.npz")


@pytest.fixture(scope="session")
def golden_smpl():
    return load_golden("smpl.npz")


@pytest.fixture(scope="session")
def golden_gcn():
    return load_golden("gcn.npz")


def sub_state_dict(g, prefix):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}
