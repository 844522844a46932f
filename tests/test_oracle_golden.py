"""Pins oracle/ (the CPU restatement) against the REAL reference's outputs.

tests/golden/reference_outputs.npz was produced by tests/golden/make_golden.py by
running /root/reference's own synergy3DMM.SynergyNet / utils.inference functions.
The tolerance here (1e-5) is 10x tighter than the 1e-4 the HIP path is held to.
"""
import os

import numpy as np
import pytest

from conftest import rel_l2, rel_max
from oracle import backbone_torch, recon_numpy, ref_loader
from synergynet_amd import synth

TOL = 1e-5


def test_backbone_oracle_matches_reference(golden, backbone_sd):
    x = synth.normalize_crops(golden['crops_u8'])
    param, pool = backbone_torch.mobilenet_v2_forward(backbone_sd, x)
    assert param.shape == (x.shape[0], 62) and pool.shape == (x.shape[0], 1280)
    assert rel_max(param.numpy(), golden['param_net']) < TOL
    assert rel_max(pool.numpy(), golden['pool_net']) < TOL


def test_batched_reconstruction_oracle_matches_reference(golden, pack):
    b = recon_numpy.Basis(pack)
    s = int(golden['vert_stride'])
    lmk = recon_numpy.reconstruct_vertex_62(b, golden['params'], dense=False)
    assert lmk.shape == (golden['params'].shape[0], 3, 68)
    assert rel_max(lmk, golden['lmk_batched']) < TOL
    lmk_nt = recon_numpy.reconstruct_vertex_62(b, golden['params'], dense=False, transform=False)
    assert rel_max(lmk_nt, golden['lmk_batched_notransform']) < TOL
    mesh = recon_numpy.reconstruct_vertex_62(b, golden['params'], dense=True)
    assert mesh.shape == (golden['params'].shape[0], 3, synth.N_VERT)
    assert rel_max(mesh[:, :, ::s], golden['mesh_batched_sub']) < TOL
    assert rel_l2(mesh.astype(np.float64).sum(axis=2), golden['mesh_batched_rowsum']) < TOL


def test_single_face_roi_and_pose_oracle_matches_reference(golden, pack):
    b = recon_numpy.Basis(pack)
    s = int(golden['vert_stride'])
    for i, (p, r) in enumerate(zip(golden['params'], golden['rois'])):
        lmk = recon_numpy.predict_vertices(b, p.copy(), list(r), dense=False)
        assert rel_max(lmk, golden['lmk_roi'][i]) < TOL
        mesh = recon_numpy.predict_vertices(b, p.copy(), list(r), dense=True)
        assert rel_max(mesh[:, ::s], golden['mesh_roi_sub'][i]) < TOL
        ang, t3d = recon_numpy.predict_pose(b, p.copy(), list(r))
        assert np.allclose(ang, golden['angles'][i], rtol=0, atol=1e-4)
        assert rel_max(t3d, golden['t3d'][i]) < TOL


def test_param_length_error(pack):
    b = recon_numpy.Basis(pack)
    with pytest.raises(RuntimeError, match='length of params mismatch'):
        recon_numpy.reconstruct_vertex_62(b, np.zeros((2, 61), np.float32))
    with pytest.raises(RuntimeError, match='length of params mismatch'):
        recon_numpy.param2vert(b, np.zeros(60, np.float32))


@pytest.mark.skipif(not ref_loader.available(), reason='/root/reference only exists in the authoring container')
def test_oracle_against_live_reference_fresh_inputs(pack, backbone_sd):
    """Fresh seeds (not the stored fixture): oracle vs the reference run live."""
    import torch
    ref, model = ref_loader.build_reference_model(pack, backbone_sd)
    x = synth.normalize_crops(synth.make_crops(2, seed=555))
    with torch.no_grad():
        want = model.forward_test(torch.from_numpy(x)).numpy()
    got, _ = backbone_torch.mobilenet_v2_forward(backbone_sd, x)
    assert rel_max(got.numpy(), want) < TOL
    params = synth.make_params(3, seed=556, scale=1.5)
    b = recon_numpy.Basis(pack)
    with torch.no_grad():
        want = model.reconstruct_vertex_62(torch.from_numpy(params), dense=True).numpy()
    assert rel_max(recon_numpy.reconstruct_vertex_62(b, params, dense=True), want) < TOL


def test_resnet50_oracle_matches_reference():
    """BASELINE config 5: oracle/resnet_torch.py vs the reference's own resnet50() module output."""
    from oracle import resnet_torch
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'resnet50_outputs.npz'))
    sd = synth.make_resnet50_state(int(g['seed']))
    x = synth.normalize_crops(synth.make_crops(2, seed=int(g['crops_seed'])))
    out, pool = resnet_torch.resnet50_forward(sd, x)
    assert out.shape == (2, 102) and pool.shape == (2, 2048)
    assert rel_max(out.numpy(), g['out102']) < TOL
    assert rel_max(pool.numpy(), g['pool']) < TOL


def test_pose_matrix_and_crop_oracles_match_reference(golden, pack):
    """predict_pose(..., ret_mat=True) and crop_img of the REAL reference (tests/golden/pose_mat_golden.npz, made by
    make_golden.py main_pose_mat) vs oracle/recon_numpy.pose_matrix and oracle/preproc_numpy.crop_img."""
    from oracle import preproc_numpy
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pose_mat_golden.npz'))
    b = recon_numpy.Basis(pack)
    got = np.stack([recon_numpy.pose_matrix(b, p.copy()) for p in golden['params']])
    assert got.shape == g['pose_mat'].shape == (golden['params'].shape[0], 3, 4)
    assert np.abs(got[:, :, :3] - g['pose_mat'][:, :, :3]).max() < 1e-6          # rotation entries are O(1)
    assert rel_max(got[:, :, 3], g['pose_mat'][:, :, 3]) < TOL
    hh, ww, seed = [int(v) for v in g['crop_frame_hw_seed']]
    frame = synth.make_frame(hh, ww, seed=seed)
    for bx, shp, s0, s1 in zip(g['crop_boxes'], g['crop_shapes'], g['crop_sums'], g['crop_weighted']):
        c = preproc_numpy.crop_img(frame, list(bx))
        assert tuple(c.shape) == tuple(shp)
        assert int(c.astype(np.int64).sum()) == int(s0)
        assert int((c.astype(np.int64).reshape(-1) * (np.arange(c.size) % 251 + 1)).sum()) == int(s1)


def test_product_lanczos_tables_equal_the_oracle_taps():
    """The tap tables the product hands to syn_crop_resize (synergynet_amd/inference.py, per-index loop) vs the oracle's
    vectorised restatement of the same published algorithm, for every crop size a detection can produce."""
    from oracle import preproc_numpy
    from synergynet_amd.inference import lanczos4_tables
    for n_src in list(range(1, 260)) + [333, 480, 719, 1024, 2047]:
        first, fixed = preproc_numpy.lanczos4_taps(120, n_src)
        x0, c = lanczos4_tables(n_src)
        assert np.array_equal(x0, first.astype(np.int32)) and np.array_equal(c, fixed.astype(np.int16)), n_src
