import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'reference_outputs.npz')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return dict(np.load(GOLDEN, allow_pickle=False))


@pytest.fixture(scope='session')
def pack(golden):
    from synergynet_amd import synth
    return synth.make_3dmm(int(golden['seeds'][1]))


@pytest.fixture(scope='session')
def backbone_sd(golden):
    from synergynet_amd import synth
    return synth.make_backbone_state(int(golden['seeds'][0]))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def rel_max(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
