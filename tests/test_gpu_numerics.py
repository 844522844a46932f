"""GPU tests of the RANGE side of the fp16x2 arithmetic (run with `-m gpu`): adversarial checkpoints / basis packs (tests/adversarial.py)
through the C ABI against the fp32 oracle, per face, at a batch of 8 (tiled kernels) and of 776 (row-marching, register-resident
and chained kernels).  The reference loads arbitrary state_dicts (synergy3DMM.py:156-164) and basis files (utils/params.py:12-35);
the tolerance is the north star's 1e-4 for the network, 2e-6 for the reconstruction alone."""
import os

import numpy as np
import pytest

import adversarial as adv

pytestmark = pytest.mark.gpu
TOL = 1e-4


def per_face_err(got, want):
    g, w = np.asarray(got, np.float64), np.asarray(want, np.float64)
    g, w = g.reshape(g.shape[0], -1), w.reshape(w.shape[0], -1)
    return np.abs(g - w).max(axis=1) / np.abs(w).max(axis=1)


def make_model(pack, sd, **env):
    from synergynet_amd.synergy3DMM import SynergyNet
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.fixture(scope='module')
def base_sd():
    from synergynet_amd import synth
    return synth.make_backbone_state(seed=31)


def run_case(pack, sd, B, expect_fallbacks):
    """Forward of B crops (degenerate images in front) on the guarded default schedule vs the oracle on a sample of faces."""
    import torch
    import warnings
    from oracle import backbone_torch
    from synergynet_amd import synth
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        model = make_model(pack, sd)
    n_fb, text = model.numerics_report()
    if expect_fallbacks is not None:
        assert (n_fb > 0) == expect_fallbacks, text
    assert bool(w) == (n_fb > 0)                                  # a fallback is announced, the default schedule is silent
    crops = adv.extreme_crops(B)
    got = model.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    pick = np.unique(np.r_[0:6, B // 2, B - 2, B - 1]) if B > 8 else np.arange(B)
    want, _ = backbone_torch.mobilenet_v2_forward(sd, synth.normalize_crops(crops[pick]))
    err = per_face_err(got[pick], want.numpy())
    return model, crops, err


CASES = {
    # block outputs driven to ~1e5 (beyond 65504) in the 64-channel 8x8 stream and the 32-channel 15x15 stream
    'stream64 x 2^14': (lambda sd: adv.scale_stream(sd, '64', 2.0 ** 14), True),
    'stream32 x 2^14': (lambda sd: adv.scale_stream(sd, '32', 2.0 ** 14), True),
    'stream160 x 2^13': (lambda sd: adv.scale_stream(sd, '160', 2.0 ** 13), True),
    # inside fp16 at scale 1, outside at the x16 of the register-resident kernels
    'stream64 x 2^8': (lambda sd: adv.scale_stream(sd, '64', 2.0 ** 8), True),
    # activations around 1e-6
    'stream64 x 2^-20': (lambda sd: adv.scale_stream(sd, '64', 2.0 ** -20), True),
    'stream24 x 2^-20': (lambda sd: adv.scale_stream(sd, '24', 2.0 ** -20), True),
    'stream160 x 2^-20': (lambda sd: adv.scale_stream(sd, '160', 2.0 ** -20), True),
    'stream64 x 2^-24': (lambda sd: adv.scale_stream(sd, '64', 2.0 ** -24), True),
    # weight rows six decades apart, no compensation anywhere
    'rows f5 1e-6..1 (BN shift kept)': (lambda sd: adv.spread_rows(sd, 5, 6.0, False), False),
    'rows f5 1e-6..1 (no shift)': (lambda sd: adv.spread_rows(sd, 5, 6.0, True), True),
    'rows f12 1e-6..1 (no shift)': (lambda sd: adv.spread_rows(sd, 12, 6.0, True), True),
    # (K = 24: the split error of 24 subnormal low pieces stays just inside the criterion -- either verdict is acceptable, the result is not)
    'rows f3 1e-6..1 (no shift)': (lambda sd: adv.spread_rows(sd, 3, 6.0, True), None),
    'rows f3 1e-8..1 (no shift)': (lambda sd: adv.spread_rows(sd, 3, 8.0, True), True),
}


@pytest.mark.parametrize('B', [8, 776])
@pytest.mark.parametrize('case', list(CASES))
def test_adversarial_checkpoint_matches_oracle(pack, base_sd, case, B):
    build, fallbacks = CASES[case]
    _, _, err = run_case(pack, build(base_sd), B, fallbacks)
    assert err.max() < TOL, f'{case} B={B}: face {err.argmax()} rel err {err.max():.3e}'


@pytest.mark.parametrize('B', [8, 1024])
def test_checkpoint_no_block_of_which_passes_the_range_proof(pack, B):
    """bench.py's `extra.all_blocks_fallback` workload (synth.make_unprovable_backbone_state: the expand rows of EVERY block spread over eight decades, BatchNorm
    shift zeroed): the load-time analysis sends all 16 blocks to their exact fp32-MFMA kernels, and what that schedule computes is the oracle's answer --
    the floor a user of an unseen checkpoint gets is a correct one (reference synergy3DMM.py:109-113, 156-164: any checkpoint loads and answers)."""
    import torch
    import warnings
    from oracle import backbone_torch
    from synergynet_amd import synth
    sd = synth.make_unprovable_backbone_state()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = make_model(pack, sd)
    n_fb, report = model.numerics_report()
    assert n_fb == 16, report
    crops = adv.extreme_crops(B)
    got = model.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    pick = np.unique(np.r_[0, 1, B // 2, B - 1])
    want, _ = backbone_torch.mobilenet_v2_forward(sd, synth.normalize_crops(crops[pick]))
    err = per_face_err(got[pick], want.numpy())
    assert err.max() < TOL, f'B={B}: face {err.argmax()} rel err {err.max():.3e}'


@pytest.mark.parametrize('B', [8, 776])
def test_the_adversarial_cases_do_break_the_unguarded_schedule(pack, base_sd, B):
    """SYNERGY_HIP_RANGE_GUARD=0 ignores the load-time verdict: the same checkpoints then run the fp16x2 kernels out of range and
    are visibly wrong -- i.e. the cases above exercise the guard, not slack in the kernels."""
    import torch
    import warnings
    from oracle import backbone_torch
    from synergynet_amd import synth
    crops = adv.extreme_crops(B)
    pick = np.arange(min(B, 6))
    for case in ('stream64 x 2^14', 'stream64 x 2^-24'):
        sd = CASES[case][0](base_sd)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            model = make_model(pack, sd, SYNERGY_HIP_RANGE_GUARD=0)
        got = model.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
        want, _ = backbone_torch.mobilenet_v2_forward(sd, synth.normalize_crops(crops[pick]))
        err = per_face_err(np.nan_to_num(got[pick], nan=1e9, posinf=1e9, neginf=-1e9), want.numpy())
        assert err.max() > 10 * TOL, f'{case}: the unguarded schedule is within {err.max():.2e} -- the case does not bite'


def test_verdict_travels_with_exported_constants(pack, base_sd):
    """A replica that imports the blob (multi-GPU hand-off) applies the sender's verdict: same bits as the sender."""
    import torch
    import warnings
    from synergynet_amd.synergy3DMM import SynergyNet
    sd = adv.scale_stream(base_sd, '64', 2.0 ** 14)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        a = make_model(pack, sd)
        b = SynergyNet(device='cuda:0', load_constants=False)
        b.import_constants(a.export_constants())
    assert b.numerics_report() == a.numerics_report() and a.numerics_report()[0] >= 4
    crops = torch.from_numpy(adv.extreme_crops(40)).cuda()
    assert torch.equal(a.forward_crops_u8(crops), b.forward_crops_u8(crops))


def test_check_numerics_self_test_without_an_oracle(pack, base_sd):
    """SynergyNet.check_numerics: default schedule vs the exact fp32-MFMA schedule of the same library on the caller's crops -- ~1e-6
    on a checkpoint inside the fp16 window, and what the unguarded schedule does to one outside of it shows up without any oracle."""
    import torch
    import warnings
    from synergynet_amd import synth
    crops = torch.from_numpy(adv.extreme_crops(24)).cuda()
    rois = torch.from_numpy(synth.make_rois(24, seed=2)).cuda()
    m = make_model(pack, base_sd)
    d = m.check_numerics(crops, rois)
    assert max(d.values()) < 2e-5, d
    assert torch.equal(m.forward_crops_u8(crops), m.forward_crops_u8(crops))          # the schedule switch was undone
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        bad = make_model(pack, adv.scale_stream(base_sd, '64', 2.0 ** 14), SYNERGY_HIP_RANGE_GUARD=0)
        good = make_model(pack, adv.scale_stream(base_sd, '64', 2.0 ** 14))
    assert bad.check_numerics(crops, rois)['param'] > 1e-3
    assert max(good.check_numerics(crops, rois).values()) < 2e-5


# ---- reconstruction: basis columns / coefficients many decades apart (one common scale lost the small columns) ----

@pytest.mark.parametrize('B', [8, 100])
def test_reconstruction_with_columns_six_decades_apart(pack, B):
    import torch
    from oracle import recon_numpy
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    wide = adv.spread_basis(pack, 3.0)
    model = SynergyNet(device='cuda:0', pack=wide, backbone_state=synth.make_backbone_state(seed=31))
    basis = recon_numpy.Basis(wide)
    param = synth.make_params(B, seed=17)
    param[0, 12:] = 0                                              # the mean face
    param[1, 12:] = 3.0                                            # every coefficient three sigma out
    alpha = param[:, 12:62] * wide['param_std'][12:62] + wide['param_mean'][12:62]
    assert np.abs(alpha).max() > 1e6 and np.abs(alpha).max() / np.abs(alpha[alpha != 0]).min() > 1e6
    roi = synth.make_rois(B, seed=5)
    pd = torch.from_numpy(param).cuda()
    for dense in (False, True):
        want = recon_numpy.reconstruct_vertex_62(basis, param, dense=dense)
        got = model.reconstruct_vertex_62(pd, dense=dense).cpu().numpy()
        e = per_face_err(got, want)
        assert e.max() < 2e-6, f'dense={dense}: face {e.argmax()} rel err {e.max():.3e}'
    want = recon_numpy.reconstruct_vertex_62(recon_numpy.Basis(pack), param, dense=True)
    got = SynergyNet(device='cuda:0', pack=pack, backbone_state=synth.make_backbone_state(seed=31)).reconstruct(pd, roi=None, dense=True).cpu().numpy()
    assert per_face_err(got, want).max() < 2e-6                    # the seeded BFM-like pack (u ~ 1e5 next to w_shp ~ 1e-3) at fp32-class accuracy


def test_reconstruction_degenerate_coefficients(pack):
    """All-zero parameters (the mean face alone), one huge coefficient, and a pack whose mean shape is zero."""
    import torch
    from oracle import recon_numpy
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    sd = synth.make_backbone_state(seed=31)
    model = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    param = np.zeros((5, 62), np.float32)
    param[1, 12 + 7] = 1.0e4                                       # one shape coefficient 1e4 sigma out
    param[2, 12:] = -50.0
    param[3] = synth.make_params(1, seed=3)[0]
    param[4, :12] = 0
    want = recon_numpy.reconstruct_vertex_62(recon_numpy.Basis(pack), param, dense=True)
    got = model.reconstruct_vertex_62(torch.from_numpy(param).cuda(), dense=True).cpu().numpy()
    assert np.isfinite(got).all() and per_face_err(got, want).max() < 2e-6
    flat = {k: np.array(v, copy=True) for k, v in pack.items()}
    flat['u_shp'][:] = 0
    flat['u_exp'][:] = 0
    m0 = SynergyNet(device='cuda:0', pack=flat, backbone_state=sd)
    want = recon_numpy.reconstruct_vertex_62(recon_numpy.Basis(flat), param[[1, 3]], dense=True)
    got = m0.reconstruct_vertex_62(torch.from_numpy(param[[1, 3]]).cuda(), dense=True).cpu().numpy()
    assert per_face_err(got, want).max() < 2e-6


def test_calibration_on_real_crops_decides_what_the_interval_estimate_cannot(pack, base_sd):
    """VERDICT r3 weak #1 / next #6b: the underflow side of the load-time analysis assumes the interval bound of a tensor is within 2^8
    of its real activations.  adv.loose_bound_stream builds a checkpoint where it is ~2^12+ looser: the static verdict accepts the
    64-channel stream (comfortable bound), its real values sit around 1e-5, and the fp16x2 kernels are visibly wrong on it.
    SynergyNet.calibrate (syn_backbone_calibrate: default vs exact schedule block by block on the caller's crops) finds the blocks
    and switches them; a healthy checkpoint is left alone."""
    import torch
    import warnings
    from oracle import backbone_torch
    from synergynet_amd import synth
    sd = adv.loose_bound_stream(base_sd)
    crops = synth.make_crops(12, seed=91)
    want, _, feats = backbone_torch.mobilenet_v2_forward(sd, synth.normalize_crops(crops), return_features=True)
    want = want.numpy()
    f7_out = [v for k, v in feats.items() if k.startswith('features.7.')][-1]        # the block output = the input of features.8
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        model = make_model(pack, sd)
    n0, text = model.numerics_report()
    # the interval bound on the input of features.8 against what the oracle really produces there: the slack the estimate cannot see
    bound8 = float([l for l in text.splitlines() if l.startswith('features.8:')][0].split('input bound ')[1].split(',')[0])
    true8 = float(f7_out.abs().max())
    print('bound / true of the input of features.8:', bound8, true8, bound8 / true8, [l for l in text.splitlines() if l.startswith('features.8:')])
    assert bound8 / true8 >= 2.0 ** 12, (bound8, true8)
    assert 'features.8:' in text and [l for l in text.splitlines() if l.startswith('features.8:')][0].endswith('-> fp16x2'), text
    cd = torch.from_numpy(crops).cuda()
    before = per_face_err(model.forward_crops_u8(cd).cpu().numpy(), want)
    print('before calibration', before.max())
    # the accepted schedule IS wrong on this checkpoint (2.2e-4 on the x16 register-resident kernels every batch size runs since the
    # small-batch chain of round 4; 1.7e-3 on the tiled kernels at input scale 1 that batches below 32 faces ran before)
    assert before.max() > 1.5 * TOL, before.max()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        n = model.calibrate(cd[:8])
    assert n >= 1 and w and 'exact schedule' in str(w[0].message)
    after = per_face_err(model.forward_crops_u8(cd).cpu().numpy(), want)
    print('switched', n, 'after calibration', after.max())
    assert after.max() < TOL, after.max()
    assert model.numerics_report()[0] == n0 + n
    # a replica that imports the calibrated constants follows the verdict
    from synergynet_amd.synergy3DMM import SynergyNet
    other = SynergyNet(device='cuda:0', load_constants=False)
    other.import_constants(model.export_constants())
    assert per_face_err(other.forward_crops_u8(cd).cpu().numpy(), want).max() < TOL
    # ... and a healthy checkpoint is left alone
    good = make_model(pack, base_sd)
    assert good.calibrate(cd[:8]) == 0 and good.numerics_report()[0] == 0


# ---- ResNet-50: no static bound (ReLU), guarded at run time ----

def test_resnet50_weight_unsafe_convolution_runs_exact_and_stays_guarded(pack):
    """one filter row 1e-25 times its neighbours fails the fp16 weight criterion -> that convolution runs the fp32-MFMA kernel.
    Its range slot used to stay 0 (= below the window) on every forward: NaN through the C ABI, a false 'activations leave the range'
    warning and an all-fp32 handle through the class.  Now the exact kernel reports into the slot like the others."""
    import torch
    import warnings
    from oracle import resnet_torch
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    sd = {k: np.array(v, copy=True) for k, v in synth.make_resnet50_state().items()}
    sd['layer2.1.conv2.weight'][5] *= np.float32(1e-25)
    crops = synth.make_crops(9, seed=4)
    x = synth.normalize_crops(crops)
    want = resnet_torch.resnet50_forward(sd, x)[0].numpy()[:, :62]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd, arch='resnet50')
        got = m.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
    n_fb, report = m.numerics_report()
    assert n_fb == 1, report                       # exactly that convolution fell back, not the handle
    assert not any('activations leave the range' in str(x.message) for x in w)
    assert m.range_status()[0] == 0 and np.isfinite(got).all()
    assert per_face_err(got, want).max() < TOL

def test_resnet50_runtime_guard(pack):
    import torch
    import warnings
    from oracle import resnet_torch
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    base = synth.make_resnet50_state()
    crops = adv.extreme_crops(12)
    x = synth.normalize_crops(crops)
    cd = torch.from_numpy(crops).cuda()
    # the seeded network stays inside the window: no fallback, no NaN
    m = SynergyNet(device='cuda:0', pack=pack, backbone_state=base, arch='resnet50')
    got = m.forward_crops_u8(cd).cpu().numpy()
    n_bad, mx = m.range_status()
    assert n_bad == 0 and 1e-3 < mx.min() and mx.max() < 6e4 and np.isfinite(got).all()
    want = resnet_torch.resnet50_forward(base, x)[0].numpy()[:, :62]
    assert per_face_err(got, want).max() < TOL
    for s in (2.0 ** 14, 2.0 ** -22):
        sd = adv.scale_resnet_stream(base, 2, s)
        want = resnet_torch.resnet50_forward(sd, x)[0].numpy()[:, :62]
        # (1) with the first-forward host check: the handle switches to the fp32-MFMA convolutions and the results are right
        m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd, arch='resnet50')
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            got = m.forward_crops_u8(cd).cpu().numpy()
        assert w and 'fp32-MFMA' in str(w[0].message)
        assert per_face_err(got, want).max() < TOL, s
        assert m.numerics_report()[0] > 0
        # (2) without it (as for any later forward): the device-side guard turns the results into NaN instead of plausible numbers
        m2 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd, arch='resnet50')
        m2._range_checked = True
        got2 = m2.forward_crops_u8(cd).cpu().numpy()
        assert np.isnan(got2).all()
        assert m2.range_status()[0] > 0
        # (3) ... and the handle recovers BY ITSELF: the poisoned forward told the library through its page-locked word, the next
        # forward runs the exact convolutions (one NaN batch, not NaN for ever); the caller never touched range_status(fallback=True)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            got3 = m2.forward_crops_u8(cd).cpu().numpy()
        assert w and 'earlier batch' in str(w[0].message)
        assert per_face_err(got3, want).max() < TOL, s
        assert m2.range_status()[0] == 0 and m2.numerics_report()[0] > 0        # guard no longer armed: nothing stale is reported
