"""AFLW2000-3D evaluator (data-gated): restatement vs the REAL reference module's outputs on seeded synthetic
ground truth (CPU), and the HIP kernel vs both (GPU)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _inputs():
    spec = importlib.util.spec_from_file_location('mg_eval', os.path.join(HERE, 'golden', 'make_golden.py'))
    src = open(os.path.join(HERE, 'golden', 'make_golden.py')).read()
    ns = {'np': np}
    start = src.index('def make_eval_inputs'); end = src.index('def main_evaluate')
    exec(src[start:end], ns)
    return ns['make_eval_inputs']()


def test_restatement_matches_reference_module_golden():
    from oracle import evaluate_numpy as ev
    fit, gt, roi, yaws = _inputs()
    g = np.load(os.path.join(HERE, 'golden', 'evaluate_golden.npz'))
    nme = ev.calc_nme(fit, gt, roi)
    assert np.array_equal(nme, g['nme'])
    np.testing.assert_allclose(ev.ana(nme, yaws), g['stats'], rtol=1e-6)


@pytest.mark.gpu
def test_device_nme_matches_reference_golden():
    from synergynet_amd import evaluate, synth
    from synergynet_amd.synergy3DMM import SynergyNet
    m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(n_vert=640), backbone_state=synth.make_backbone_state())
    fit, gt, roi, yaws = _inputs()
    g = np.load(os.path.join(HERE, 'golden', 'evaluate_golden.npz'))
    nme = evaluate.calc_nme(m, fit, gt, roi)
    np.testing.assert_allclose(nme, g['nme'], rtol=2e-7, atol=0)          # one float32 ulp (mean summation order)
    np.testing.assert_allclose(evaluate.ana(nme, yaws), g['stats'], rtol=1e-5)
    # params -> landmarks -> statistics and the pose error run end to end
    p = synth.make_params(96, seed=4)
    stats = evaluate.benchmark_aflw2000_params(m, p, gt, roi, yaws)
    assert len(stats) == 5 and all(np.isfinite(stats))
    mae = evaluate.benchmark_FOE(m, p, np.zeros((94, 3)), [3, 7])
    assert len(mae) == 4 and all(np.isfinite(mae))
