"""Seeded synthetic assets for the SynergyNet hot path.

The reference's real assets (3dmm_data/*.npy, param_whitening.pkl, tri.mat,
pretrained/best.pth.tar) are Google-Drive downloads that are not in the
reference checkout (reference README.md:54-59, utils/params.py:12-24,
synergy3DMM.py:73-77).  Parity and benchmarks therefore run on synthetic
assets with the same shapes, dtypes and roughly the same magnitudes.

Everything here is numpy-only and driven by numpy's PCG64 streams, so the same
seed gives bit-identical assets on every machine (the GPU box regenerates the
assets instead of shipping 40 MB of fixtures).
"""
from __future__ import annotations

import numpy as np

N_VERT = 53215          # reference synergy3DMM.py:135 (.view(-1, 53215, 3))
N_LMK = 68              # reference synergy3DMM.py:116 (lmk_pts=68)
N_SHP, N_EXP = 40, 10   # reference synergy3DMM.py:35-36
STD_SIZE = 120          # reference utils/params.py:34
N_TRI = 105840          # triangles [3,105840]

# MobileNetV2 (width 1.0) inverted-residual table: t, c, n, s
# (reference backbone_nets/mobilenetv2_backbone.py:107-117)
MBV2_CFG = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
            (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


def mbv2_layers():
    """Ordered conv-layer table of the reference backbone.

    Each entry: dict(key=<state_dict prefix of the conv weight>, bn=<prefix of
    its BatchNorm>, kind='stem'|'pw'|'dw', cin, cout, stride, relu6, feature=<index
    into .features>, residual=<True on the project conv of a residual block>).
    Key names follow reference mobilenetv2_backbone.py:33-74,131-143.
    """
    layers = [dict(key='features.0.0', bn='features.0.1', kind='stem', cin=3, cout=32,
                   stride=2, relu6=True, feature=0, residual=False)]
    inp, f = 32, 1
    for t, c, n, s in MBV2_CFG:
        for i in range(n):
            stride = s if i == 0 else 1
            hid = inp * t
            res = (stride == 1 and inp == c)
            j = 0
            if t != 1:
                layers.append(dict(key=f'features.{f}.conv.0.0', bn=f'features.{f}.conv.0.1', kind='pw',
                                   cin=inp, cout=hid, stride=1, relu6=True, feature=f, residual=False))
                j = 1
            layers.append(dict(key=f'features.{f}.conv.{j}.0', bn=f'features.{f}.conv.{j}.1', kind='dw',
                               cin=hid, cout=hid, stride=stride, relu6=True, feature=f, residual=False))
            layers.append(dict(key=f'features.{f}.conv.{j + 1}', bn=f'features.{f}.conv.{j + 2}', kind='pw',
                               cin=hid, cout=c, stride=1, relu6=False, feature=f, residual=res))
            inp = c
            f += 1
    layers.append(dict(key='features.18.0', bn='features.18.1', kind='pw', cin=320, cout=1280,
                       stride=1, relu6=True, feature=18, residual=False))
    return layers


HEADS = [('classifier_ori.1', 12), ('classifier_shape.1', 40), ('classifier_exp.1', 10)]


def make_backbone_state(seed: int = 1234) -> dict:
    """Backbone state_dict (numpy float32 arrays, reference key names, no prefix).

    Conv weights are variance-preserving (kaiming fan_in) rather than the
    reference's fan_out init (mobilenetv2_backbone.py:161-171): with fan_out the
    depthwise layers shrink the signal by ~1/C and the output stops depending on
    the image, which would hide early-layer bugs from the parity tests.  BN
    running statistics are randomised so scale/shift handling is exercised.
    """
    rng = np.random.default_rng(seed)
    sd = {}
    for L in mbv2_layers():
        cin, cout = L['cin'], L['cout']
        if L['kind'] == 'dw':
            shape, fan_in = (cout, 1, 3, 3), 9
        elif L['kind'] == 'stem':
            shape, fan_in = (cout, cin, 3, 3), cin * 9
        else:
            shape, fan_in = (cout, cin, 1, 1), cin
        gain = 2.0 if L['relu6'] else 1.0
        sd[L['key'] + '.weight'] = (rng.standard_normal(shape) * np.sqrt(gain / fan_in)).astype(np.float32)
        sd[L['bn'] + '.weight'] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
        sd[L['bn'] + '.bias'] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        sd[L['bn'] + '.running_mean'] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        sd[L['bn'] + '.running_var'] = rng.uniform(0.8, 1.25, cout).astype(np.float32)
        sd[L['bn'] + '.num_batches_tracked'] = np.array(1000, dtype=np.int64)
    for name, n in HEADS:
        sd[name + '.weight'] = (rng.standard_normal((n, 1280)) * 0.02).astype(np.float32)
        sd[name + '.bias'] = (rng.standard_normal(n) * 0.1).astype(np.float32)
    return sd


def resnet50_convs():
    """Ordered conv table of the reference ResNet-50 (backbone_nets/resnet_backbone.py:139-254, Bottleneck :90-136):
    dicts with key (conv weight prefix), bn, cin, cout, k, stride, relu, plus block bookkeeping."""
    convs = [dict(key='conv1', bn='bn1', cin=3, cout=64, k=7, stride=2, relu=True, role='stem')]
    inpl = 64
    for L, (planes, nblk) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for i in range(nblk):
            stride = 2 if (i == 0 and L > 1) else 1
            pre = f'layer{L}.{i}'
            convs.append(dict(key=pre + '.conv1', bn=pre + '.bn1', cin=inpl, cout=planes, k=1, stride=1, relu=True, role='c1', block=pre))
            convs.append(dict(key=pre + '.conv2', bn=pre + '.bn2', cin=planes, cout=planes, k=3, stride=stride, relu=True, role='c2', block=pre))
            convs.append(dict(key=pre + '.conv3', bn=pre + '.bn3', cin=planes, cout=planes * 4, k=1, stride=1, relu=False, role='c3', block=pre))
            if i == 0:
                convs.append(dict(key=pre + '.downsample.0', bn=pre + '.downsample.1', cin=inpl, cout=planes * 4, k=1,
                                  stride=stride, relu=False, role='ds', block=pre))
            inpl = planes * 4
    return convs


RESNET_HEADS = [('fc_tex', 40), ('fc_ori', 12), ('fc_shape', 40), ('fc_exp', 10)]      # module order (:185-188)


def make_resnet50_state(seed: int = 2468) -> dict:
    """ResNet-50 state_dict (numpy fp32, reference key names).  Variance-preserving conv init, randomised BN
    statistics; the last BN of every residual branch gets a small gamma so 16 stacked blocks stay bounded."""
    rng = np.random.default_rng(seed)
    sd = {}
    for c in resnet50_convs():
        fan_in = c['cin'] * c['k'] * c['k']
        gain = 2.0 if c['relu'] else 1.0
        sd[c['key'] + '.weight'] = (rng.standard_normal((c['cout'], c['cin'], c['k'], c['k'])) * np.sqrt(gain / fan_in)).astype(np.float32)
        lo, hi = (0.2, 0.4) if c['role'] == 'c3' else (0.8, 1.2)
        sd[c['bn'] + '.weight'] = rng.uniform(lo, hi, c['cout']).astype(np.float32)
        sd[c['bn'] + '.bias'] = (rng.standard_normal(c['cout']) * 0.1).astype(np.float32)
        sd[c['bn'] + '.running_mean'] = (rng.standard_normal(c['cout']) * 0.1).astype(np.float32)
        sd[c['bn'] + '.running_var'] = rng.uniform(0.8, 1.25, c['cout']).astype(np.float32)
        sd[c['bn'] + '.num_batches_tracked'] = np.array(1000, dtype=np.int64)
    for name, n in RESNET_HEADS:
        sd[name + '.weight'] = (rng.standard_normal((n, 2048)) * 0.02).astype(np.float32)
        sd[name + '.bias'] = (rng.standard_normal(n) * 0.1).astype(np.float32)
    return sd


def _rotation(yaw, pitch, roll):
    cy, sy = np.cos(yaw), np.sin(yaw)
    cp, sp = np.cos(pitch), np.sin(pitch)
    cr, sr = np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Rz @ Rx @ Ry


def make_3dmm(seed: int = 4321, n_vert: int = N_VERT) -> dict:
    """Synthetic 3DMM pack with the reference's file-level fields.

    Returns the arrays the reference loads from 3dmm_data/ (utils/params.py:12-24):
    keypoints [204] int64 (flat indices [3k,3k+1,3k+2], utils/io.py:78-81),
    w_shp [3N,40], w_exp [3N,10], u_shp [3N,1], u_exp [3N,1], param_mean [62],
    param_std [62] (all float32), tri [3,N_TRI] int32 1-based.  Magnitudes mimic
    3DDFA's BFM pack: vertices ~1e4-1e5 model units, scale ~5e-4, so projected
    faces land inside the 120x120 crop.
    """
    rng = np.random.default_rng(seed)
    n3 = 3 * n_vert
    u = np.empty((n_vert, 3), dtype=np.float64)
    # points on a bumpy half-ellipsoid, roughly face sized in BFM units
    th = rng.uniform(-1.2, 1.2, n_vert)
    ph = rng.uniform(-1.3, 1.3, n_vert)
    r = 1.0 + 0.05 * rng.standard_normal(n_vert)
    u[:, 0] = 8.0e4 * r * np.sin(th) * np.cos(ph * 0.5)
    u[:, 1] = 9.5e4 * r * np.sin(ph)
    u[:, 2] = 6.0e4 * r * np.cos(th) * np.cos(ph) - 1.0e4
    u = u.reshape(n3, 1)
    u_exp = (rng.standard_normal((n3, 1)) * 300.0)
    u_shp = u - u_exp
    w_shp = rng.standard_normal((n3, N_SHP)) / np.sqrt(n3)
    w_shp *= np.linspace(1.0, 0.25, N_SHP)[None, :]
    w_exp = rng.standard_normal((n3, N_EXP)) * 400.0 * np.linspace(1.0, 0.3, N_EXP)[None, :]

    R = _rotation(0.35, -0.15, 0.08)
    s = 5.2e-4
    Pm = np.concatenate([s * R, np.array([[61.0], [58.5], [-40.0]])], axis=1)   # rows of [P|t]
    param_mean = np.concatenate([Pm.reshape(-1),
                                 rng.standard_normal(N_SHP) * 6.0e4,
                                 rng.standard_normal(N_EXP) * 0.4])
    P_std = np.concatenate([np.full((3, 3), 1.1e-4), np.array([[9.0], [11.0], [25.0]])], axis=1)
    param_std = np.concatenate([P_std.reshape(-1),
                                rng.uniform(3.0e4, 3.0e5, N_SHP),
                                rng.uniform(0.3, 1.5, N_EXP)])

    kp_vert = np.sort(rng.choice(n_vert, N_LMK, replace=False))
    keypoints = np.stack([3 * kp_vert, 3 * kp_vert + 1, 3 * kp_vert + 2], axis=1).reshape(-1).astype(np.int64)
    tri = (rng.integers(0, n_vert, size=(3, N_TRI)) + 1).astype(np.int32)
    return dict(keypoints=keypoints,
                w_shp=w_shp.astype(np.float32), w_exp=w_exp.astype(np.float32),
                u_shp=u_shp.astype(np.float32), u_exp=u_exp.astype(np.float32),
                param_mean=param_mean.astype(np.float32), param_std=param_std.astype(np.float32),
                tri=tri)


def make_crops(batch: int, seed: int = 99, smooth: bool = False) -> np.ndarray:
    """uint8 BGR crops [B,120,120,3] (what cv2.resize hands to synergy3DMM.py:188-189)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(batch, STD_SIZE, STD_SIZE, 3), dtype=np.uint8)
    if smooth:
        f = img.astype(np.float32)
        for _ in range(3):
            f = (f + np.roll(f, 1, 1) + np.roll(f, -1, 1) + np.roll(f, 1, 2) + np.roll(f, -1, 2)) / 5.0
        f = (f - f.mean()) * 4.0 + 127.5
        img = np.clip(np.rint(f), 0, 255).astype(np.uint8)
    return img


def normalize_crops(img_u8: np.ndarray) -> np.ndarray:
    """HWC uint8 -> NCHW float32 (x-127.5)/128 (reference synergy3DMM.py:189-192)."""
    x = img_u8.astype(np.float32).transpose(0, 3, 1, 2)
    return np.ascontiguousarray((x - np.float32(127.5)) / np.float32(128.0))


def make_params(batch: int, seed: int = 7, scale: float = 1.0) -> np.ndarray:
    """Whitened 62-d parameter vectors ~N(0, scale) as the backbone would emit them."""
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((batch, 62)) * scale).astype(np.float32)


def make_rois(batch: int, seed: int = 11) -> np.ndarray:
    """ROI boxes [B,5] = sx, sy, ex, ey, score with side 80-400 px."""
    rng = np.random.default_rng(seed)
    side = rng.uniform(80, 400, batch)
    sx = rng.uniform(0, 600, batch)
    sy = rng.uniform(0, 400, batch)
    # non-square boxes exercise scale_x != scale_y (utils/inference.py:129-136)
    ex = sx + side * rng.uniform(0.9, 1.1, batch)
    ey = sy + side
    return np.stack([sx, sy, ex, ey, rng.uniform(0.9, 1.0, batch)], axis=1).astype(np.float32)


def make_grid_topology(rows: int = 231, cols: int = 231, n_vert: int | None = None, n_tri: int | None = None) -> np.ndarray:
    """Triangles [ntri,3] int32 (0-based, counter-clockwise in image coordinates) of a rows x cols vertex grid, two per
    cell -- the BFM mesh is such a parametrised surface (53215 vertices, 105840 triangles ~ a 231 x 231 grid).
    n_vert drops the trailing vertices (and the triangles that use them); n_tri pads by repeating leading triangles
    (duplicates exercise the rasteriser's first-wins tie rule) or truncates."""
    r, c = np.meshgrid(np.arange(rows - 1), np.arange(cols - 1), indexing='ij')
    v00 = (r * cols + c).reshape(-1)
    v01, v10, v11 = v00 + 1, v00 + cols, v00 + cols + 1
    tri = np.stack([np.stack([v00, v10, v01], 1), np.stack([v01, v10, v11], 1)], 1).reshape(-1, 3)
    if n_vert is not None:
        tri = tri[(tri < n_vert).all(1)]
    if n_tri is not None:
        if tri.shape[0] >= n_tri:
            tri = tri[:n_tri]
        else:
            tri = np.concatenate([tri, tri[:n_tri - tri.shape[0]]], 0)
    return np.ascontiguousarray(tri, dtype=np.int32)


def make_face_meshes(n_faces: int, rows: int = 231, cols: int = 231, n_vert: int | None = None, height: int = 450,
                     width: int = 450, seed: int = 777) -> np.ndarray:
    """[F,3,N] float32 meshes in IMAGE coordinates (x, y, depth) -- the layout syn_reconstruct writes: bumpy half-ellipsoids
    on the grid of make_grid_topology, each face with its own centre, size and in-plane rotation."""
    rng = np.random.default_rng(seed)
    n = rows * cols
    v, u = np.meshgrid(np.linspace(-1, 1, rows), np.linspace(-1, 1, cols), indexing='ij')
    u, v = u.reshape(-1), v.reshape(-1)
    out = np.empty((n_faces, 3, n), dtype=np.float64)
    for f in range(n_faces):
        cx, cy = rng.uniform(0.3, 0.7) * width, rng.uniform(0.3, 0.7) * height
        rx, ry = rng.uniform(0.12, 0.3) * width, rng.uniform(0.15, 0.33) * height
        a = rng.uniform(-0.4, 0.4)
        jx, jy = rng.standard_normal(n) * 0.15, rng.standard_normal(n) * 0.15
        x0, y0 = rx * u + jx, ry * v + jy
        out[f, 0] = cx + np.cos(a) * x0 - np.sin(a) * y0
        out[f, 1] = cy + np.sin(a) * x0 + np.cos(a) * y0
        out[f, 2] = 0.8 * rx * np.cos(0.5 * np.pi * u) * np.cos(0.5 * np.pi * v) + rng.standard_normal(n) * 0.3 - 20.0 * f
    if n_vert is not None:
        out = out[:, :, :n_vert]
    return np.ascontiguousarray(out, dtype=np.float32)


# ---- FaceBoxes detector: layer table shared by the oracle, the packer and the tests ----
def faceboxes_convs():
    """(state_dict prefix, cin, cout, k, stride, pad, kind) for every convolution of FaceBoxesNet in forward order
    (reference FaceBoxes/models/faceboxes.py:64-150).  kind: 'crelu' (conv+BN, cat[x,-x], ReLU :48-61), 'bn' (conv+BN+ReLU
    :8-18), 'head' (conv with bias, no activation :104-114)."""
    L = [('conv1', 3, 24, 7, 4, 3, 'crelu'), ('conv2', 48, 64, 5, 2, 2, 'crelu')]
    for i in (1, 2, 3):
        p = f'inception{i}.'
        L += [(p + 'branch1x1', 128, 32, 1, 1, 0, 'bn'), (p + 'branch1x1_2', 128, 32, 1, 1, 0, 'bn'),
              (p + 'branch3x3_reduce', 128, 24, 1, 1, 0, 'bn'), (p + 'branch3x3', 24, 32, 3, 1, 1, 'bn'),
              (p + 'branch3x3_reduce_2', 128, 24, 1, 1, 0, 'bn'), (p + 'branch3x3_2', 24, 32, 3, 1, 1, 'bn'),
              (p + 'branch3x3_3', 32, 32, 3, 1, 1, 'bn')]
    L += [('conv3_1', 128, 128, 1, 1, 0, 'bn'), ('conv3_2', 128, 256, 3, 2, 1, 'bn'),
          ('conv4_1', 256, 128, 1, 1, 0, 'bn'), ('conv4_2', 128, 256, 3, 2, 1, 'bn')]
    for i, (cin, a) in enumerate(((128, 21), (256, 1), (256, 1))):
        L += [(f'loc.{i}', cin, a * 4, 3, 1, 1, 'head'), (f'conf.{i}', cin, a * 2, 3, 1, 1, 'head')]
    return L


def make_faceboxes_state(seed: int = 1357) -> dict:
    """Seeded FaceBoxesNet state_dict (numpy): variance-preserving conv weights, randomised BN statistics, head biases tuned
    so that a few hundred priors pass the 0.05 confidence threshold and a handful pass 0.5 on noise images."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, cin, cout, k, _, _, kind in faceboxes_convs():
        fan = cin * k * k
        w = rng.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / fan)
        if name == 'conv1':
            w /= 60.0                                            # pixels minus mean are O(100): bring activations to O(1)
        if kind == 'head':
            sd[name + '.weight'] = (w * (1.2 if name.startswith('conf') else 0.25)).astype(np.float32)
            b = rng.standard_normal(cout) * 0.1
            if name.startswith('conf'):
                b[0::2] += 4.0                                   # background logit ahead: most priors are negatives
            sd[name + '.bias'] = b.astype(np.float32)
        else:
            sd[name + '.conv.weight'] = w.astype(np.float32)
            sd[name + '.bn.weight'] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
            sd[name + '.bn.bias'] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            sd[name + '.bn.running_mean'] = (rng.standard_normal(cout) * 0.1).astype(np.float32)
            sd[name + '.bn.running_var'] = rng.uniform(0.8, 1.2, cout).astype(np.float32)
    return sd


def make_frame(height: int, width: int, seed: int = 11) -> np.ndarray:
    """uint8 BGR frame [H,W,3]: low-pass noise with a few bright blobs."""
    rng = np.random.default_rng(seed)
    f = rng.uniform(0, 255, (height, width, 3)).astype(np.float32)
    for _ in range(2):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5.0
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(4):
        cy, cx, r = rng.uniform(0.2, 0.8) * height, rng.uniform(0.2, 0.8) * width, rng.uniform(0.05, 0.15) * min(height, width)
        f += 90.0 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * r * r))[:, :, None]
    return np.clip(np.rint(f), 0, 255).astype(np.uint8)


def make_unprovable_backbone_state(seed: int = 1234, decades: float = 8.0) -> dict:
    """A MobileNetV2 checkpoint none of whose blocks passes the load-time range proof of the fp16x2 kernels: the output rows
    of every block's first convolution (the expand 1x1; features.1: the depthwise-free project) are spread over `decades` orders of
    magnitude with their BatchNorm shift zeroed, so the small rows lose their low fp16 piece to subnormals and the weight criterion
    sends the block to the exact fp32-MFMA kernel.  A valid network (it computes something else than make_backbone_state's): what a user
    of an unseen checkpoint gets AT WORST -- bench.py times it as `extra.all_blocks_fallback`."""
    sd = {k: np.array(v, copy=True) for k, v in make_backbone_state(seed).items()}
    rng = np.random.default_rng(seed + 77)
    first = {}
    for L in mbv2_layers():
        if L['kind'] == 'pw' and 2 <= L['feature'] <= 17 and L['feature'] not in first:
            first[L['feature']] = L
    for f, L in first.items():
        n = sd[L['key'] + '.weight'].shape[0]
        sc = (10.0 ** rng.uniform(-decades, 0.0, n)).astype(np.float32)
        sc[rng.integers(0, n)] = 1.0
        sd[L['key'] + '.weight'] = sd[L['key'] + '.weight'] * sc[:, None, None, None]
        sd[L['bn'] + '.bias'] = np.zeros(n, np.float32)
        sd[L['bn'] + '.running_mean'] = np.zeros(n, np.float32)
    return sd
