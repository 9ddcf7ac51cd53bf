import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
SDSS_SIGMA = np.array([0.873, 0.348, 0.418, 0.873, 3.476])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def golden():
    return load_golden


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
