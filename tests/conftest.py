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


ASSETS = os.path.join(ROOT, 'tests', 'golden', '_assets')


def real_detector_assets():
    """(weights path, {i: BGR uint8 frame}) of the reference's trained FaceBoxes weights and sample photographs, or None.
    They are staged by tests/golden/make_golden.py / __graft_entry__.build() into the git-ignored tests/golden/_assets/ (the
    weights may also come from $SYN_FACEBOXES_WEIGHTS); absent -> the real-data tests skip."""
    w = os.environ.get('SYN_FACEBOXES_WEIGHTS') or os.path.join(ASSETS, 'FaceBoxesProd.pth')
    if not os.path.isfile(w) or not all(os.path.isfile(os.path.join(ASSETS, f'sample_{i}.jpg')) for i in range(1, 5)):
        return None
    from PIL import Image
    frames = {i: np.ascontiguousarray(np.asarray(Image.open(os.path.join(ASSETS, f'sample_{i}.jpg')).convert('RGB'))[:, :, ::-1])
              for i in range(1, 5)}
    return w, frames
