"""GPU parity at the FULL sizes of BASELINE.json's configs, on DISTINCT faces, straight against the oracle (run with `-m gpu`).

configs[1] B = 128, configs[2]/[3] B = 1024 (per GPU), configs[4] ResNet-50 B = 512 -- plus B = 512 on MobileNetV2.  Every face is
a different seeded crop (white-noise and low-pass halves), so a data-dependent or tile-position-dependent fault of the large-batch
schedule (persistent workgroups of features.2-7, un-sliced features.15-17 + head, uint8 stem on the matrix pipe) cannot hide
behind repeated inputs.  The whole chain is compared per face: uint8 crops -> parameters -> 68 landmarks -> 53215-vertex mesh
(row-pitched default output and the reference's packed layout) -> pose, HIP path (C ABI) vs oracle/backbone_torch.py +
oracle/recon_numpy.py (pinned to the real reference by tests/test_oracle_golden.py).  Tolerance 1e-4 relative per face
(BASELINE.json north_star); angles 1e-3 degrees.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


def per_face_rel(got, want):
    B = want.shape[0]
    g, w = np.asarray(got, np.float64).reshape(B, -1), np.asarray(want, np.float64).reshape(B, -1)
    return np.abs(g - w).max(axis=1) / np.maximum(np.abs(w).max(axis=1), 1e-30)


def distinct_crops(B, seed):
    from synergynet_amd import synth
    c = synth.make_crops(B, seed=seed)
    c[B // 2:] = synth.make_crops(B - B // 2, seed=seed + 1, smooth=True)
    assert len({c[i].tobytes() for i in range(B)}) == B
    return c


@pytest.fixture(scope='module')
def model(pack, backbone_sd):
    import torch
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from synergynet_amd.synergy3DMM import SynergyNet
    return SynergyNet(device='cuda:0', pack=pack, backbone_state=backbone_sd)       # default schedule (fused, fp16x2 operands)


@pytest.fixture(scope='module')
def basis(pack):
    from oracle import recon_numpy
    return recon_numpy.Basis(pack)


def oracle_params(sd, crops, chunk=128):
    from oracle import backbone_torch
    from synergynet_amd import synth
    ps, pools = [], []
    for i in range(0, crops.shape[0], chunk):
        p, pool = backbone_torch.mobilenet_v2_forward(sd, synth.normalize_crops(crops[i:i + chunk]))
        ps.append(p.numpy()); pools.append(pool.numpy())
    return np.concatenate(ps), np.concatenate(pools)


@pytest.mark.parametrize('B', [128, 512, 1024])
def test_distinct_faces_whole_chain_matches_oracle(model, basis, backbone_sd, B):
    import torch
    from oracle import recon_numpy
    from synergynet_amd import synth
    crops = distinct_crops(B, seed=7000 + B)
    rois = synth.make_rois(B, seed=7100 + B)
    want_p, want_pool = oracle_params(backbone_sd, crops)

    cd = torch.from_numpy(crops).cuda()
    param, pool = model.forward_crops_u8(cd, return_pool=True)
    e = per_face_rel(param.cpu().numpy(), want_p)
    assert e.max() < TOL, f'parameters: worst face {e.argmax()} rel err {e.max():.3e}'
    e = per_face_rel(pool.cpu().numpy(), want_pool)
    assert e.max() < TOL, f'pooled feature: worst face {e.argmax()} rel err {e.max():.3e}'
    # fp32 NCHW ingest (the reference's forward_test signature) on the same faces
    pf = model.forward_test(torch.from_numpy(synth.normalize_crops(crops)).cuda())
    e = per_face_rel(pf.cpu().numpy(), want_p)
    assert e.max() < TOL, f'forward_test: worst face {e.argmax()} rel err {e.max():.3e}'

    # 68 landmarks, batched method (no ROI affine): every face
    lmk = model.reconstruct(param, dense=False).cpu().numpy()
    want_l = recon_numpy.reconstruct_vertex_62(basis, want_p, dense=False)
    e = per_face_rel(lmk, want_l)
    assert e.max() < TOL, f'landmarks: worst face {e.argmax()} rel err {e.max():.3e}'

    # 53215-vertex mesh, batched method: every face, default (row-pitched) output and the reference's packed layout
    mesh = model.reconstruct(param, dense=True)
    assert tuple(mesh.shape) == (B, 3, synth.N_VERT) and mesh.stride(1) % 128 == 0
    packed = torch.empty((B, 3, synth.N_VERT), dtype=torch.float32, device='cuda')
    model.reconstruct(param, dense=True, out=packed)
    assert torch.equal(packed, mesh)
    worst = 0.0
    for i in range(0, B, 128):
        want_m = recon_numpy.reconstruct_vertex_62(basis, want_p[i:i + 128], dense=True)
        e = per_face_rel(mesh[i:i + 128].cpu().numpy(), want_m)
        assert e.max() < TOL, f'mesh: worst face {i + e.argmax()} rel err {e.max():.3e}'
        worst = max(worst, float(e.max()))
    del packed

    # ROI affine (utils/inference.py:127-138) + pose (:146-157): landmarks and pose on every face, dense on every 41st face
    lmk_r = model.reconstruct(param, roi=rois, dense=False).cpu().numpy()
    mesh_r = model.reconstruct(param, roi=rois, dense=True)
    ang, t3d = model.predict_pose_batch(param, rois)
    ang, t3d = ang.cpu().numpy(), t3d.cpu().numpy()
    for i in range(B):
        wl = recon_numpy.predict_vertices(basis, want_p[i].copy(), list(rois[i]), dense=False)
        assert per_face_rel(lmk_r[i:i + 1], wl[None]).max() < TOL, f'ROI landmarks, face {i}'
        wa, wt = recon_numpy.predict_pose(basis, want_p[i].copy(), list(rois[i]))
        assert np.allclose(ang[i], wa, rtol=0, atol=1e-3), f'pose angles, face {i}: {ang[i]} vs {wa}'
        assert per_face_rel(t3d[i:i + 1], np.asarray(wt)[None]).max() < TOL, f't3d, face {i}'
    for i in range(0, B, 41):
        wm = recon_numpy.predict_vertices(basis, want_p[i].copy(), list(rois[i]), dense=True)
        assert per_face_rel(mesh_r[i:i + 1].cpu().numpy(), wm[None]).max() < TOL, f'ROI mesh, face {i}'
    print(f'B={B}: worst per-face mesh rel err {worst:.3e}')


@pytest.mark.parametrize('B', [480, 515, 1024])
def test_fp32_crops_take_the_row_marching_stem_and_match_oracle(model, backbone_sd, B):
    """forward_test (reference synergy3DMM.py:151-154: normalised fp32 NCHW crops) at batches that fill the chip runs the F32
    instantiation of the row-marching stem (stem_rm.hip, round 4: both operands as two fp16 pieces) instead of the tiled bf16 x3 kernel:
    every face against the oracle on distinct faces, and against the uint8 ingest of the same pixels (another stem arithmetic: 1e-5)."""
    import torch
    from oracle import backbone_torch
    from synergynet_amd import synth
    crops = synth.make_crops(B, seed=4242)
    crops[: B // 2] = synth.make_crops(B // 2, seed=4243, smooth=True)
    crops[0] = 0; crops[1] = 255                                   # flat images: the padding value must be exact
    x = synth.normalize_crops(crops)
    got = model.forward_test(torch.from_numpy(x).cuda()).cpu().numpy()
    via_u8 = model.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
    assert np.isfinite(got).all()
    assert (np.abs(got - via_u8).max(axis=1) / np.abs(via_u8).max(axis=1)).max() < 1e-5
    idx = np.r_[0:8, B // 2 - 3:B // 2 + 3, B - 9:B]               # the oracle on the first / middle / last faces (first and last workgroup rounds)
    want = backbone_torch.mobilenet_v2_forward(backbone_sd, x[idx])[0].numpy()
    assert (np.abs(got[idx] - want).max(axis=1) / np.abs(want).max(axis=1)).max() < 1e-4
    # values far outside [-1, 1] but inside the documented domain (|x| < 6e4): still the fp32-class result of the same linear map + ReLU6
    xs = (x[:B] * np.float32(100.0)).astype(np.float32)
    got2 = model.forward_test(torch.from_numpy(xs).cuda()).cpu().numpy()
    want2 = backbone_torch.mobilenet_v2_forward(backbone_sd, xs[idx])[0].numpy()
    assert (np.abs(got2[idx] - want2).max(axis=1) / np.abs(want2).max(axis=1)).max() < 1e-4


def test_resnet50_b512_distinct_faces_match_oracle(pack, basis):
    """BASELINE configs[4]: ResNet-50, B = 512 distinct faces + the full mesh, per face vs the oracle."""
    import torch
    from oracle import recon_numpy, resnet_torch
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    sd = synth.make_resnet50_state(2468)
    m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd, arch='resnet50')
    B = 512
    crops = distinct_crops(B, seed=7600)
    wp, wpool = [], []
    for i in range(0, B, 64):
        p, pool = resnet_torch.resnet50_forward(sd, synth.normalize_crops(crops[i:i + 64]))
        wp.append(p.numpy()[:, :62]); wpool.append(pool.numpy())
    want_p, want_pool = np.concatenate(wp), np.concatenate(wpool)
    param, pool = m.forward_crops_u8(torch.from_numpy(crops).cuda(), return_pool=True)
    e = per_face_rel(param.cpu().numpy(), want_p)
    assert e.max() < TOL, f'parameters: worst face {e.argmax()} rel err {e.max():.3e}'
    e = per_face_rel(pool.cpu().numpy(), want_pool)
    assert e.max() < TOL, f'pooled feature: worst face {e.argmax()} rel err {e.max():.3e}'
    mesh = m.reconstruct(param, dense=True)
    for i in range(0, B, 128):
        want_m = recon_numpy.reconstruct_vertex_62(basis, want_p[i:i + 128], dense=True)
        e = per_face_rel(mesh[i:i + 128].cpu().numpy(), want_m)
        assert e.max() < TOL, f'mesh: worst face {i + e.argmax()} rel err {e.max():.3e}'


def test_resnet50_beyond_the_32bit_offset_window(pack):
    """ResNet-50 at B = 2400: layer 1's tensors are 2.2 GB, past what the buffer-load kernels address with 32-bit byte offsets
    (conv_lt_kernel / conv_h2s_kernel: tensors < 2 GiB, csrc/resnet_kernels.hip launch_conv_f16x2) -- those convolutions must fall back to
    the flat-addressed kernel while the smaller deep-layer tensors stay on the LDS-tiled one.  The batch is 300 copies of 8 distinct
    faces; every copy must reproduce a B = 136 run of the same faces (fused pairs on both sides; the generic kernel walks K in the same
    order: equal to fp32 rounding) and the copies must agree with each other bit for bit."""
    import torch
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import SynergyNet
    sd = synth.make_resnet50_state(2468)
    m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd, arch='resnet50')
    eight = distinct_crops(8, seed=4242)
    small = m.forward_crops_u8(torch.from_numpy(np.tile(eight, (17, 1, 1, 1))).cuda())[:8].cpu().numpy()
    big = m.forward_crops_u8(torch.from_numpy(np.tile(eight, (300, 1, 1, 1))).cuda())
    assert tuple(big.shape)[0] == 2400 and torch.isfinite(big).all()
    bigv = big.view(300, 8, -1)
    assert torch.equal(bigv[0], bigv[137]) and torch.equal(bigv[0], bigv[299])
    e = per_face_rel(bigv[0].cpu().numpy(), small)
    assert e.max() < 1e-5, f'face {e.argmax()}: {e.max():.3e}'
    assert m.range_status()[0] == 0


def test_pose_matrix_matches_reference_golden_and_oracle(model, basis, golden):
    """predict_pose(..., ret_mat=True) (utils/inference.py:146-157): vs the REAL reference's output (pose_mat_golden.npz) and,
    on a big batch of distinct parameter vectors, vs the oracle."""
    import os
    from oracle import recon_numpy
    from synergynet_amd import inference, synth
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'pose_mat_golden.npz'))
    got = model.pose_matrix_batch(golden['params']).cpu().numpy()
    assert got.shape == g['pose_mat'].shape
    assert np.abs(got[:, :, :3] - g['pose_mat'][:, :, :3]).max() < 1e-5
    assert per_face_rel(got[:, :, 3], g['pose_mat'][:, :, 3]).max() < TOL
    P = inference.predict_pose(golden['params'][1], list(golden['rois'][1]), ret_mat=True)      # the reference's module-level name
    assert isinstance(P, np.ndarray) and P.shape == (3, 4) and P.dtype == np.float32 and np.array_equal(P, got[1])
    params = synth.make_params(777, seed=31, scale=1.2)
    got = model.pose_matrix_batch(params).cpu().numpy()
    want = np.stack([recon_numpy.pose_matrix(basis, p.copy()) for p in params])
    assert np.abs(got[:, :, :3] - want[:, :, :3]).max() < 1e-5
    assert per_face_rel(got[:, :, 3], want[:, :, 3]).max() < TOL
    with pytest.raises(RuntimeError, match='length of params mismatch'):
        model.predict_pose_batch(np.zeros((3, 61), np.float32))
    with pytest.raises(RuntimeError, match=r'roi must be \[B,5\]'):
        model.predict_pose_batch(np.zeros((3, 62), np.float32), np.zeros((3, 4), np.float32))


def test_batch_beyond_every_baseline_size(model, basis, backbone_sd):
    """B = 4100 faces in ONE call (4x the per-GPU shard of configs[3], not a multiple of anything): index arithmetic, persistent-loop
    rounds and workspace growth beyond the sizes the kernels were tuned at.  A sample of faces from every region of the batch vs the
    oracle (parameters, landmarks, pitched mesh), the mesh's landmark columns vs the landmark launch, and finiteness of everything."""
    import torch
    from oracle import recon_numpy
    from synergynet_amd import synth
    B = 4100
    crops = synth.make_crops(B, seed=8100)
    crops[1::2] = synth.make_crops(B // 2, seed=8101, smooth=True)
    rois = synth.make_rois(B, seed=8102)
    cd, rd = torch.from_numpy(crops).cuda(), torch.from_numpy(rois).cuda()
    param = model.forward_crops_u8(cd)
    lmk = model.reconstruct(param, roi=rd, dense=False)
    mesh = model.reconstruct(param, roi=rd, dense=True)
    ang, t3d = model.predict_pose_batch(param, rd)
    torch.cuda.synchronize()
    assert torch.isfinite(param).all() and torch.isfinite(lmk).all() and torch.isfinite(mesh).all() and torch.isfinite(ang).all()
    pick = np.unique(np.r_[0:4, 1023:1027, 2047:2051, 3071:3075, 4093:4100])
    want_p, _ = oracle_params(backbone_sd, crops[pick])
    e = per_face_rel(param[torch.from_numpy(pick).cuda()].cpu().numpy(), want_p)
    assert e.max() < TOL, f'parameters: face {pick[e.argmax()]} rel err {e.max():.3e}'
    for i in pick:
        got_p = param[i].cpu().numpy()
        wl = recon_numpy.predict_vertices(basis, got_p, rois[i], dense=False, transform=True)
        wm = recon_numpy.predict_vertices(basis, got_p, rois[i], dense=True, transform=True)
        assert per_face_rel(lmk[i].cpu().numpy()[None], wl[None]).max() < 1e-5, i
        assert per_face_rel(mesh[i].cpu().numpy()[None], wm[None]).max() < 1e-5, i
    # every face: the landmarks are the keypoint columns of its mesh (the same contraction through two launches)
    kp = torch.as_tensor(np.asarray(basis.keypoints[::3] // 3)).cuda()
    d = (mesh[:, :, kp] - lmk).abs().amax() / lmk.abs().amax()
    assert float(d) < 1e-5
    # and the first 1024 faces equal a 1024-face call bit for bit (same kernels from B = 768 on; position independence)
    assert torch.equal(model.forward_crops_u8(cd[:1024]), param[:1024])
