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

  * backbone_nets.resnet_backbone.resnet50        -> resnet50_outputs.npz   (main_resnet50)
  * Sim3DR.RenderPipeline / get_normal / rasterize: the reference's own Python package driving its own C++ rasteriser
    (compiled from where it lies by oracle/Makefile)  -> render_golden.npz  (main_render)
  * FaceBoxes.models.faceboxes.FaceBoxesNet, utils.prior_box.PriorBox, utils.box_utils.decode, glued like
    FaceBoxes.__call__                              -> faceboxes_golden.npz (main_faceboxes)
  * benchmark_aflw2000.calc_nme / ana on seeded synthetic ground truth -> evaluate_golden.npz (main_evaluate)

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


def main_pose_mat():
    """predict_pose(..., ret_mat=True) of the reference (utils/inference.py:146-157) on the parameter vectors / boxes of
    reference_outputs.npz -> tests/golden/pose_mat_golden.npz.  Also records the reference's own crop_img on boxes that
    overhang every border of a seeded frame (utils/inference.py:95-125) as a checksum per crop."""
    g = dict(np.load(os.path.join(HERE, 'reference_outputs.npz')))
    pack = synth.make_3dmm(int(g['seeds'][1]))
    sd = synth.make_backbone_state(int(g['seeds'][0]))
    ref_loader.build_reference_model(pack, sd)
    inf = ref_loader._REF_MODULES['utils.inference']
    mats = np.stack([np.asarray(inf.predict_pose(p.copy(), list(r), ret_mat=True), dtype=np.float32) for p, r in zip(g['params'], g['rois'])])
    frame = synth.make_frame(97, 131, seed=12)
    boxes = np.array([[-20.4, -10.6, 60.2, 70.5, 1], [90.5, 50.5, 160.4, 120.6, 1], [10, 20, 50, 60, 1], [-5.5, 80.5, 140.5, 110.4, 1],
                      [40.49, -30.51, 80.5, 9.5, 1]], dtype=np.float64)
    crops = [inf.crop_img(frame, list(bx)) for bx in boxes]
    fp = os.path.join(HERE, 'pose_mat_golden.npz')
    np.savez_compressed(fp, pose_mat=mats, crop_frame_hw_seed=np.array([97, 131, 12]), crop_boxes=boxes,
                        crop_shapes=np.array([c.shape for c in crops]), crop_sums=np.array([int(c.astype(np.int64).sum()) for c in crops]),
                        crop_weighted=np.array([int((c.astype(np.int64).reshape(-1) * (np.arange(c.size) % 251 + 1)).sum()) for c in crops]))
    print('wrote', fp, os.path.getsize(fp), 'bytes; P[0] =', mats[0])


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


def load_reference_sim3dr():
    """The REAL reference Sim3DR package (Sim3DR/Sim3DR.py, Sim3DR/lighting.py) imported from /root/reference, with its
    Cython extension `Sim3DR_Cython` (not built here) replaced by ctypes calls into the reference's own C++ compiled by
    oracle/Makefile (oracle/_ref/libsim3dr_ref.so).  Nothing is copied."""
    import ctypes as C
    import importlib
    import types
    from oracle import sim3dr as osim
    osim.build()
    lib = C.CDLL(os.path.join(os.path.dirname(osim.__file__), '_ref', 'libsim3dr_ref.so'))
    lib.ref_get_normal.argtypes = [C.c_void_p] * 3 + [C.c_int] * 2
    lib.ref_rasterize.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_int]
    stub = types.ModuleType('Sim3DR_Cython')

    def get_normal(normal, vertices, triangles, nver, ntri):                    # rasterize.pyx:66-74
        assert normal.dtype == np.float32 and vertices.dtype == np.float32 and triangles.dtype == np.int32
        lib.ref_get_normal(normal.ctypes.data, vertices.ctypes.data, triangles.ctypes.data, nver, ntri)

    def rasterize(image, vertices, triangles, colors, depth_buffer, ntri, h, w, c, alpha=1, reverse=False):   # rasterize.pyx:96-110
        assert image.dtype == np.uint8 and vertices.dtype == np.float32 and triangles.dtype == np.int32 and colors.dtype == np.float32
        lib.ref_rasterize(image.ctypes.data, vertices.ctypes.data, triangles.ctypes.data, colors.ctypes.data,
                          depth_buffer.ctypes.data, ntri, h, w, c, alpha, int(reverse))

    stub.get_normal, stub.rasterize = get_normal, rasterize
    sys.modules['Sim3DR_Cython'] = stub
    sys.path.insert(0, ref_loader.REF_ROOT)
    try:
        for k in [k for k in sys.modules if k == 'Sim3DR' or k.startswith('Sim3DR.')]:
            del sys.modules[k]
        return importlib.import_module('Sim3DR')
    finally:
        sys.path.remove(ref_loader.REF_ROOT)


def main_render():
    """the reference's own RenderPipeline (Sim3DR/lighting.py) + C++ rasteriser on seeded grid meshes with the
    configuration of utils/render.py:18-27 -> tests/golden/render_golden.npz."""
    from oracle import sim3dr as osim
    ref = load_reference_sim3dr()
    out = {}
    for name, rows, cols, nv, nt, hw, nf in (('small', 40, 44, None, None, 160, 3), ('full', 231, 231, 53215, 105840, 450, 2)):
        tri = synth.make_grid_topology(rows, cols, n_vert=nv, n_tri=nt)
        meshes = synth.make_face_meshes(nf, rows, cols, n_vert=nv, height=hw, width=hw, seed=900 + rows)
        img = np.random.default_rng(rows).integers(0, 256, (hw, hw, 3), dtype=np.uint8)
        app = ref.RenderPipeline(**osim.RENDER_CFG)
        overlap = img.copy()
        normals, lights = [], []
        for f in range(nf):
            ver = np.ascontiguousarray(meshes[f].T)
            normals.append(ref.get_normal(ver, tri))
            overlap = app(ver, tri, overlap)
        if name == 'small':
            # vertex colours of the reference pipeline: re-run its lighting code path with a rasterize probe
            captured = []
            real_rasterize = sys.modules['Sim3DR.lighting'].rasterize
            sys.modules['Sim3DR.lighting'].rasterize = lambda v, t, c, bg=None, **kw: (captured.append(c.copy()), bg)[1]
            for f in range(nf):
                app(np.ascontiguousarray(meshes[f].T), tri, img.copy())
            sys.modules['Sim3DR.lighting'].rasterize = real_rasterize
            out['small_normal'] = np.stack(normals)
            out['small_light'] = np.stack(captured)
            out['small_img'] = img
        out[name + '_overlay'] = overlap
        out[name + '_cfg'] = np.array([rows, cols, nv or rows * cols, nt or tri.shape[0], hw, nf, 900 + rows, rows], dtype=np.int64)
    path = os.path.join(HERE, 'render_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, 'KB')


def main_faceboxes():
    """the reference's own FaceBoxesNet / PriorBox / decode (torch + numpy only, importable as they are) and
    its pure-python NMS with the comparison of the Cython variant the wrapper dispatches to, glued exactly like
    FaceBoxes.__call__ (FaceBoxes.py:60-143) on seeded weights and frames -> tests/golden/faceboxes_golden.npz.
    Frames stay below 720x1080 here: the down-scaling branch needs cv2.resize, which is absent (restated, unpinned)."""
    import importlib
    import types
    fbroot = os.path.join(ref_loader.REF_ROOT, 'FaceBoxes')
    # import the sub-modules without running FaceBoxes/__init__.py (which imports cv2 and the Cython NMS)
    pkg = types.ModuleType('FaceBoxes'); pkg.__path__ = [fbroot]; sys.modules['FaceBoxes'] = pkg
    for sub in ('models', 'utils'):
        m = types.ModuleType('FaceBoxes.' + sub); m.__path__ = [os.path.join(fbroot, sub)]; sys.modules['FaceBoxes.' + sub] = m
    net_mod = importlib.import_module('FaceBoxes.models.faceboxes')
    prior_mod = importlib.import_module('FaceBoxes.utils.prior_box')
    box_mod = importlib.import_module('FaceBoxes.utils.box_utils')
    cfg = importlib.import_module('FaceBoxes.utils.config').cfg
    sd = synth.make_faceboxes_state()
    net = net_mod.FaceBoxesNet(phase='test', size=None, num_classes=2)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)      # num_batches_tracked stays default
    net.eval()
    from oracle import faceboxes_torch as ofb
    out = {}
    for tag, (hh, ww) in (('a', (300, 420)), ('b', (97, 131)), ('c', (64, 64))):
        frame = synth.make_frame(hh, ww, seed=hh)
        img = np.float32(frame)
        scale_bbox = torch.Tensor([img.shape[1], img.shape[0], img.shape[1], img.shape[0]])
        img -= (104, 117, 123)
        with torch.no_grad():
            loc, conf = net(torch.from_numpy(img.transpose(2, 0, 1)).unsqueeze(0))
        priors = prior_mod.PriorBox(image_size=(hh, ww)).forward()
        boxes = box_mod.decode(loc.data.squeeze(0), priors.data, cfg['variance'])
        boxes = (boxes * scale_bbox / 1 / 1).cpu().numpy()
        scores = conf.squeeze(0).data.cpu().numpy()[:, 1]
        inds = np.where(scores > 0.05)[0]
        boxes, scores = boxes[inds], scores[inds]
        order = scores.argsort()[::-1][:5000]
        dets = np.hstack((boxes[order], scores[order][:, np.newaxis])).astype(np.float32, copy=False)
        keep = ofb.cpu_nms(dets, 0.3) if dets.shape[0] else []      # cpu_nms.pyx semantics (restated: the Cython module is not built)
        out[tag + '_loc'] = loc.numpy()[0]
        out[tag + '_conf'] = conf.numpy()[0]
        out[tag + '_priors'] = priors.numpy()
        out[tag + '_dets'] = dets[keep, :][:750]
        out[tag + '_hw'] = np.array([hh, ww])
    path = os.path.join(HERE, 'faceboxes_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items()}, os.path.getsize(path) // 1024, 'KB')
    for k in [k for k in sys.modules if k == 'FaceBoxes' or k.startswith('FaceBoxes.')]:
        del sys.modules[k]


ASSETS = os.path.join(HERE, '_assets')          # git-ignored; travels to the GPU box with the snapshot like oracle/_ref


def stage_real_detector_assets():
    """Copies the reference's trained detector weights (FaceBoxes/weights/FaceBoxesProd.pth, 4 MB) and its sample photographs
    (img/sample_*.jpg) into tests/golden/_assets/ -- data files, not source; the directory is git-ignored so they stay out of
    history, but it is not gpurun-ignored, so the real-data GPU test can run on the GPU box (where /root/reference is absent)."""
    import shutil
    if not os.path.isfile(os.path.join(ref_loader.REF_ROOT, 'FaceBoxes', 'weights', 'FaceBoxesProd.pth')):
        return False
    os.makedirs(ASSETS, exist_ok=True)
    shutil.copyfile(os.path.join(ref_loader.REF_ROOT, 'FaceBoxes', 'weights', 'FaceBoxesProd.pth'), os.path.join(ASSETS, 'FaceBoxesProd.pth'))
    for i in range(1, 5):
        shutil.copyfile(os.path.join(ref_loader.REF_ROOT, 'img', f'sample_{i}.jpg'), os.path.join(ASSETS, f'sample_{i}.jpg'))
    return True


def load_bgr(path):
    """cv2.imread stand-in: PIL decode (libjpeg), RGB -> BGR.  The decoded frame's SHA-256 is stored in the fixture so a test on
    another machine knows whether it is looking at the same pixels."""
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(path).convert('RGB'))[:, :, ::-1])


def main_faceboxes_real():
    """The reference's own FaceBoxesNet with its TRAINED weights (FaceBoxes/weights/FaceBoxesProd.pth) on its own sample
    photographs (img/sample_*.jpg), glued exactly like FaceBoxes.__call__ (FaceBoxes.py:60-143) -> faceboxes_real_golden.npz:
    realistic candidate counts and real faces for the device detector.  Frames above 720x1080 take the down-scaling branch,
    whose cv2.resize is absent here: those frames are down-scaled with the oracle's restatement (unpinned) and flagged
    `scaled` in the fixture; samples 2 and 3 need no scaling and are the reference end to end."""
    import hashlib
    import importlib
    import types
    assert stage_real_detector_assets(), 'reference detector weights not found'
    fbroot = os.path.join(ref_loader.REF_ROOT, 'FaceBoxes')
    pkg = types.ModuleType('FaceBoxes'); pkg.__path__ = [fbroot]; sys.modules['FaceBoxes'] = pkg
    for sub in ('models', 'utils'):
        m = types.ModuleType('FaceBoxes.' + sub); m.__path__ = [os.path.join(fbroot, sub)]; sys.modules['FaceBoxes.' + sub] = m
    net_mod = importlib.import_module('FaceBoxes.models.faceboxes')
    prior_mod = importlib.import_module('FaceBoxes.utils.prior_box')
    box_mod = importlib.import_module('FaceBoxes.utils.box_utils')
    cfg = importlib.import_module('FaceBoxes.utils.config').cfg
    net = net_mod.FaceBoxesNet(phase='test', size=None, num_classes=2)
    ck = torch.load(os.path.join(ASSETS, 'FaceBoxesProd.pth'), map_location='cpu')
    net.load_state_dict({(k.split('module.', 1)[-1] if k.startswith('module.') else k): v for k, v in ck.items()}, strict=False)   # utils/functions.py:21-25
    net.eval()
    from oracle import faceboxes_torch as ofb
    out = {}
    for i in range(1, 5):
        frame = load_bgr(os.path.join(ASSETS, f'sample_{i}.jpg'))
        h, w = frame.shape[:2]
        scale = 1                                            # FaceBoxes.py:63-80
        if h > 720:
            scale = 720 / h
        if w * scale > 1080:
            scale *= 1080 / (w * scale)
        small = frame if scale == 1 else ofb.resize_linear_u8(frame, int(scale * h), int(scale * w))
        img = np.float32(small)
        im_height, im_width, _ = img.shape
        scale_bbox = torch.Tensor([img.shape[1], img.shape[0], img.shape[1], img.shape[0]])
        img -= (104, 117, 123)
        with torch.no_grad():
            loc, conf = net(torch.from_numpy(img.transpose(2, 0, 1)).unsqueeze(0))
        priors = prior_mod.PriorBox(image_size=(im_height, im_width)).forward()
        boxes = box_mod.decode(loc.data.squeeze(0), priors.data, cfg['variance'])
        boxes = (boxes * scale_bbox / scale / 1).cpu().numpy()
        scores = conf.squeeze(0).data.cpu().numpy()[:, 1]
        inds = np.where(scores > 0.05)[0]
        n_cand = inds.size
        boxes, scores = boxes[inds], scores[inds]
        order = scores.argsort()[::-1][:5000]
        dets = np.hstack((boxes[order], scores[order][:, np.newaxis])).astype(np.float32, copy=False)
        keep = ofb.cpu_nms(dets, 0.3) if dets.shape[0] else []
        dets = dets[keep, :][:750]
        tag = f's{i}'
        out[tag + '_hw'] = np.array([h, w]); out[tag + '_scaled'] = np.array(scale != 1)
        out[tag + '_sha256'] = np.array(hashlib.sha256(frame.tobytes()).hexdigest())
        out[tag + '_loc_sub'] = loc.numpy()[0][::7].copy(); out[tag + '_conf_sub'] = conf.numpy()[0][::7].copy()
        out[tag + '_loc_absmax'] = np.array(np.abs(loc.numpy()).max())
        out[tag + '_n_cand'] = np.array(n_cand)
        out[tag + '_dets'] = dets
        print(tag, (h, w), 'scale', scale, 'priors', loc.shape[1], 'candidates', n_cand, 'dets', dets.shape[0], 'faces (score > 0.5):', int((dets[:, 4] > 0.5).sum()))
    path = os.path.join(HERE, 'faceboxes_real_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KB')
    for k in [k for k in sys.modules if k == 'FaceBoxes' or k.startswith('FaceBoxes.')]:
        del sys.modules[k]


def make_eval_inputs(n=96, seed=2024):
    """Seeded stand-ins for aflw2000_data/eval/*.npy (absent): ground-truth landmarks, crop boxes, yaws, fitted landmarks."""
    rng = np.random.default_rng(seed)
    roi = np.stack([rng.uniform(0, 200, n), rng.uniform(0, 200, n)], 1)
    side = rng.uniform(80, 400, n)
    roi = np.concatenate([roi, roi + side[:, None]], 1).astype(np.float32)                      # sx, sy, ex, ey
    fit = rng.uniform(10, 110, (n, 2, 68)).astype(np.float32)                                   # crop coordinates
    sx, sy, ex, ey = roi.T
    gt = np.stack([fit[:, 0] * ((ex - sx) / 120)[:, None] + sx[:, None], fit[:, 1] * ((ey - sy) / 120)[:, None] + sy[:, None],
                   rng.uniform(-50, 50, (n, 68))], 1)
    gt[:, :2] += rng.standard_normal((n, 2, 68)) * 3.0
    yaws = rng.uniform(-90, 90, n).astype(np.float32)
    return fit, gt.astype(np.float32), roi, yaws


def main_evaluate():
    """SURVEY 8f row 4 (data-gated part): the reference's own benchmark_aflw2000.calc_nme / ana with its `_load` patched to
    serve seeded synthetic ground truth -> tests/golden/evaluate_golden.npz."""
    import importlib
    import types
    fit, gt, roi, yaws = make_eval_inputs()
    served = {'AFLW2000-3D.pose.npy': yaws, 'AFLW2000-3D.pts68.npy': gt, 'AFLW2000-3D-Reannotated.pts68.npy': gt[:, :, ::-1].copy(),
              'AFLW2000-3D_crop.roi_box.npy': roi}
    io_stub = types.ModuleType('utils.io'); io_stub._load = lambda fp: served[os.path.basename(fp)]
    utils_stub = types.ModuleType('utils'); utils_stub.__path__ = []; utils_stub.io = io_stub
    saved = {k: sys.modules.get(k) for k in ('utils', 'utils.io', 'benchmark_aflw2000')}
    sys.modules['utils'], sys.modules['utils.io'] = utils_stub, io_stub
    sys.modules.pop('benchmark_aflw2000', None)
    sys.path.insert(0, ref_loader.REF_ROOT)
    try:
        ref = importlib.import_module('benchmark_aflw2000')
        nme = ref.calc_nme([f.copy() for f in fit], option='ori')
        import contextlib, io as _io
        with contextlib.redirect_stdout(_io.StringIO()):
            stats = ref.ana(nme)
    finally:
        sys.path.remove(ref_loader.REF_ROOT)
        for k, v in saved.items():
            if v is None: sys.modules.pop(k, None)
            else: sys.modules[k] = v
    path = os.path.join(HERE, 'evaluate_golden.npz')
    np.savez_compressed(path, nme=nme, stats=np.array(stats, dtype=np.float64))
    print('wrote', path, nme.shape, stats)


def main_write_obj():
    """The reference's own write_obj (utils/inference.py:8-23) on a seeded mesh: the bytes of the file it writes."""
    import tempfile
    pack = synth.make_3dmm(SEED_3DMM)
    sd = synth.make_backbone_state(SEED_W)
    ref_loader.build_reference_model(pack, sd)
    inf = ref_loader._REF_MODULES['utils.inference']
    rng = np.random.default_rng(77)
    nv, nt = 2500, 4100
    # magnitudes of a posed mesh in image space, values that round at the fourth decimal, negative zero, exact halves
    vert = (rng.standard_normal((3, nv)) * np.array([[60.0], [60.0], [40.0]]) + np.array([[60.0], [60.0], [0.0]])).astype(np.float32)
    vert[:, :6] = np.array([[0.00005, -0.00005, 0.12345, -0.0, 1e-7, 119.99995], [0.5, -2.5, 1234.56785, 0.0, -1e-7, 3.00005],
                            [1e5, -1e5, 0.99995, -0.99995, 2.00015, 7.0]], dtype=np.float32)
    tri = rng.integers(1, nv + 1, size=(3, nt)).astype(np.int32)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, arg in (('plain', 'mesh'), ('with_ext', 'mesh2.obj')):
            cwd = os.getcwd()
            os.chdir(td)
            try:
                inf.write_obj(arg, vert, tri)
            finally:
                os.chdir(cwd)
            fn = arg if arg.endswith('.obj') else arg + '.obj'
            out['bytes_' + name] = np.frombuffer(open(os.path.join(td, fn), 'rb').read(), dtype=np.uint8)
            out['name_' + name] = np.array(fn)
    path = os.path.join(HERE, 'write_obj_golden.npz')
    np.savez_compressed(path, vertices=vert, triangles=tri, **out)
    print('wrote', path, out['bytes_plain'].size, 'bytes of OBJ')


if __name__ == '__main__':
    if len(sys.argv) > 1:                      # python make_golden.py main_pose_mat main_faceboxes_real ...: only those fixtures
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    main()
    main_pose_mat()
    main_resnet50()
    main_render()
    main_faceboxes()
    main_faceboxes_real()
    main_evaluate()
    main_write_obj()
