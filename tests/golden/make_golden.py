"""Generate tests/golden/*.npz by running the REAL reference code (authoring container only).

    python tests/golden/make_golden.py

Imports /root/reference through oracle/ref_loader.py (stubs for the absent
torchvision/cv2/FaceBoxes, synthetic 3dmm_data served in place of the absent
files) and records what the reference's own functions return on seeded
synthetic inputs:

  * SynergyNet.forward_test                      (synergy3DMM.py:151-154)
  * SynergyNet.reconstruct_vertex_62 sparse/dense (synergy3DMM.py:116-149)
  * utils.inference.predict_sparseVert / predict_denseVert / predict_pose
                                                 (utils/inference.py:127-157)

The assets themselves (9 MB of weights, 32 MB of basis) are NOT stored: they are
regenerated bit-identically from the seeds in synergynet_amd/synth.py.  The dense
mesh is stored as a strided vertex subset plus float64 per-row sums.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))

from synergynet_amd import synth          # noqa: E402
from oracle import ref_loader             # noqa: E402

SEED_W, SEED_3DMM, SEED_IMG, SEED_PARAM, SEED_ROI = 1234, 4321, 99, 7, 11
B_NET = 4            # faces through the backbone
B_PARAM = 6          # random whitened parameter vectors straight into reconstruction
VERT_STRIDE = 53     # dense-mesh subset: vertices 0, 53, 106, ...


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    pack = synth.make_3dmm(SEED_3DMM)
    sd = synth.make_backbone_state(SEED_W)
    ref, model = ref_loader.build_reference_model(pack, sd)
    inf = ref_loader._REF_MODULES['utils.inference']

    crops = synth.make_crops(B_NET, SEED_IMG)
    crops[B_NET // 2:] = synth.make_crops(B_NET - B_NET // 2, SEED_IMG + 1, smooth=True)
    x = torch.from_numpy(synth.normalize_crops(crops))
    with torch.no_grad():
        param_net = model.forward_test(x)                       # [B,62]
        _, pool = model.I2P.forward_test(x)                     # pooled 1280-d feature
    params = np.concatenate([param_net.numpy(), synth.make_params(B_PARAM, SEED_PARAM)], axis=0).astype(np.float32)
    rois = synth.make_rois(params.shape[0], SEED_ROI)

    pt = torch.from_numpy(params)
    with torch.no_grad():
        lmk_b = model.reconstruct_vertex_62(pt, dense=False).numpy()           # [B,3,68]
        mesh_b = model.reconstruct_vertex_62(pt, dense=True).numpy()           # [B,3,53215]
        lmk_b_nt = model.reconstruct_vertex_62(pt, dense=False, transform=False).numpy()

    lmk_roi, mesh_roi, angles, t3d = [], [], [], []
    for p, r in zip(params, rois):
        lmk_roi.append(inf.predict_sparseVert(p.copy(), list(r), transform=True))
        mesh_roi.append(inf.predict_denseVert(p.copy(), list(r), transform=True))
        a, t = inf.predict_pose(p.copy(), list(r))
        angles.append([float(v) for v in a])
        t3d.append(np.asarray(t, dtype=np.float32))
    mesh_roi = np.stack(mesh_roi).astype(np.float32)

    out = dict(
        seeds=np.array([SEED_W, SEED_3DMM, SEED_IMG, SEED_PARAM, SEED_ROI]),
        b_net=np.array(B_NET), vert_stride=np.array(VERT_STRIDE),
        crops_u8=crops, param_net=param_net.numpy(), pool_net=pool.numpy(),
        params=params, rois=rois,
        lmk_batched=lmk_b, lmk_batched_notransform=lmk_b_nt,
        mesh_batched_sub=mesh_b[:, :, ::VERT_STRIDE].copy(), mesh_batched_rowsum=mesh_b.astype(np.float64).sum(axis=2),
        lmk_roi=np.stack(lmk_roi).astype(np.float32),
        mesh_roi_sub=mesh_roi[:, :, ::VERT_STRIDE].copy(), mesh_roi_rowsum=mesh_roi.astype(np.float64).sum(axis=2),
        angles=np.asarray(angles, dtype=np.float64), t3d=np.stack(t3d),
        torch_version=np.array(torch.__version__), numpy_version=np.array(np.__version__),
    )
    fp = os.path.join(HERE, 'reference_outputs.npz')
    np.savez_compressed(fp, **out)
    print('wrote', fp, os.path.getsize(fp), 'bytes')
    print('param_net[0,:6]', param_net.numpy()[0, :6])
    print('lmk range', lmk_b.min(), lmk_b.max(), 'mesh range', mesh_b.min(), mesh_b.max())
    print('angles', np.asarray(angles)[:3])

    # sensitivity check: the output must depend on the image (early-layer bugs visible)
    with torch.no_grad():
        x2 = x.clone()
        x2[:, :, :8, :8] += 0.25
        d = (model.forward_test(x2) - param_net).abs().max().item()
    print('max |d param| for a corner perturbation:', d)


def main_resnet50():
    """BASELINE config 5: outputs of the reference's own resnet50() module (backbone_nets/resnet_backbone.py:304-312,
    importable as-is: torch-only) on seeded crops -> tests/golden/resnet50_outputs.npz."""
    import importlib
    sd = synth.make_resnet50_state(2468)
    sys.path.insert(0, ref_loader.REF_ROOT)
    try:
        m = importlib.import_module('backbone_nets.resnet_backbone')
    finally:
        sys.path.remove(ref_loader.REF_ROOT)
    net = m.resnet50().eval()
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = synth.normalize_crops(synth.make_crops(2, seed=31))
    feats = {}
    net.avgpool.register_forward_hook(lambda mod, i, o: feats.__setitem__('pool', o.flatten(1)))
    with torch.no_grad():
        out = net(torch.from_numpy(x)).numpy()
    fp = os.path.join(HERE, 'resnet50_outputs.npz')
    np.savez_compressed(fp, seed=np.array(2468), crops_seed=np.array(31), out102=out, pool=feats['pool'].numpy())
    print('wrote', fp, os.path.getsize(fp), 'bytes; out[0,:6] =', out[0, :6])


if __name__ == '__main__':
    main()
    main_resnet50()
