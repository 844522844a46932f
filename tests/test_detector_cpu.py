"""CPU tests of the FaceBoxes oracle against fixtures produced by the REAL reference modules
(tests/golden/make_golden.py main_faceboxes: FaceBoxesNet, PriorBox, decode imported from /root/reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import faceboxes_torch as ofb
from synergynet_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def fgold():
    return dict(np.load(os.path.join(HERE, 'golden', 'faceboxes_golden.npz')))


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_oracle_matches_reference_golden(fgold, tag):
    sd = synth.make_faceboxes_state()
    hh, ww = [int(v) for v in fgold[tag + '_hw']]
    frame = synth.make_frame(hh, ww, seed=hh)
    img = np.float32(frame) - np.array((104, 117, 123), dtype=np.float32)
    loc, conf = ofb.net_forward(sd, torch.from_numpy(img.transpose(2, 0, 1)).unsqueeze(0))
    np.testing.assert_allclose(loc.numpy()[0], fgold[tag + '_loc'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(conf.numpy()[0], fgold[tag + '_conf'], rtol=1e-5, atol=1e-6)
    assert np.array_equal(ofb.prior_boxes((hh, ww)).numpy(), fgold[tag + '_priors'])
    dets = ofb.detect(sd, frame, return_all=True)
    assert dets.shape == fgold[tag + '_dets'].shape
    np.testing.assert_allclose(dets, fgold[tag + '_dets'], rtol=1e-5, atol=1e-3)


def test_large_frames_are_scaled_like_the_wrapper():
    """FaceBoxes.py:63-80: frames above 720x1080 are shrunk first and the boxes mapped back by 1/scale."""
    sd = synth.make_faceboxes_state()
    frame = synth.make_frame(800, 1300, seed=3)
    dets = ofb.detect(sd, frame, return_all=True)
    assert dets.shape[1] == 5 and dets.shape[0] > 0 and dets[:, 2].max() > 1080            # boxes are in original-frame pixels
    small = ofb.resize_linear_u8(frame, 664, 1080)
    assert small.shape == (664, 1080, 3) and abs(float(small.mean()) - float(frame.mean())) < 1.0


def test_nms_follows_the_cython_variant():
    d = np.array([[0, 0, 10, 10, 0.9], [0, 0, 10, 10, 0.8], [20, 20, 30, 30, 0.7], [1, 1, 11, 11, 0.6]], dtype=np.float32)
    assert ofb.cpu_nms(d, 0.3) == [0, 2]
    assert ofb.cpu_nms(d[:0], 0.3) == []


def test_oracle_detector_with_trained_weights_on_real_photographs():
    """oracle/faceboxes_torch.py vs the REAL reference FaceBoxesNet carrying its TRAINED weights on its own sample photographs
    (tests/golden/faceboxes_real_golden.npz, made by make_golden.py main_faceboxes_real): realistic candidate counts, real faces."""
    import hashlib
    import os
    import torch
    from conftest import real_detector_assets
    from oracle import faceboxes_torch as ofb
    assets = real_detector_assets()
    if assets is None:
        pytest.skip('tests/golden/_assets (reference detector weights + sample photographs) not staged')
    wpath, frames = assets
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'faceboxes_real_golden.npz'))
    ck = torch.load(wpath, map_location='cpu')
    sd = {(k.split('module.', 1)[-1] if k.startswith('module.') else k): v.numpy() for k, v in ck.items()}
    for i in (2, 3, 1):
        if hashlib.sha256(frames[i].tobytes()).hexdigest() != str(g[f's{i}_sha256']):
            pytest.skip('this machine decodes the JPEGs to different pixels than the authoring container')
        got = ofb.detect(sd, frames[i], return_all=True)
        want = g[f's{i}_dets']
        assert got.shape == want.shape
        np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=0, atol=1e-5)
        np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=0, atol=2e-2)
    assert int((g['s1_dets'][:, 4] > 0.5).sum()) == 10          # the group photograph
