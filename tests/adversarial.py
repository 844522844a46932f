"""Adversarial checkpoints / 3DMM packs for the range side of the fp16x2 arithmetic.

The reference loads arbitrary state_dicts and basis files (reference synergy3DMM.py:156-164, utils/params.py:12-35); the default
kernels carry every GEMM operand as two fp16 pieces, whose exponent range is 2^-14 .. 65504.  These builders move tensors of a
seeded network out of that window WITHOUT changing what the network computes (exact power-of-two rescaling of a residual stream,
compensated in the consumers' weights), or put rows / columns of one tensor many decades apart, so that the HIP path can be held
to the fp32 oracle on them.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

from synergynet_amd import synth

# residual groups of the MobileNetV2 table (reference mobilenetv2_backbone.py:107-117): the tensor that flows from `first - 1`'s
# project into blocks first .. last (residual adds) and into `last + 1`'s expand
GROUPS = {'24': (3, 3), '32': (5, 6), '64': (8, 10), '96': (12, 13), '160': (15, 16)}


def _block_keys(f):
    L = [l for l in synth.mbv2_layers() if l['feature'] == f]
    expand = L[0] if len(L) == 3 else None
    return expand, L[-2], L[-1]


def scale_stream(sd: dict, group: str, s: float) -> dict:
    """The residual stream of `group` times s (a power of two: every step below is exact in fp32, so the network's outputs do not
    change by a bit -- only the magnitude of the tensors at the block boundaries first-1 | first | ... | last | last+1 does).
      * project BN of blocks first-1 .. last: gamma, beta x s   (their contribution to the stream)
      * expand conv of blocks first .. last+1: weights / s      (hidden activations unchanged)"""
    first, last = GROUPS[group]
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    for f in range(first - 1, last + 1):
        _, _, proj = _block_keys(f)
        out[proj['bn'] + '.weight'] = out[proj['bn'] + '.weight'] * np.float32(s)
        out[proj['bn'] + '.bias'] = out[proj['bn'] + '.bias'] * np.float32(s)
    for f in range(first, last + 2):
        expand, _, _ = _block_keys(f)
        out[expand['key'] + '.weight'] = out[expand['key'] + '.weight'] * np.float32(1.0 / s)
    return out


def loose_bound_stream(sd: dict, s: float = 2.0 ** -15, A: float = 24.0) -> dict:
    """The 64-channel stream times s (scale_stream) AND a static bound on it that is ~2^12 looser than what features.7 really
    produces: hidden channel 2i+1 of features.7 becomes an exact copy of channel 2i (expand, depthwise and their BatchNorms), and the
    project weights of the pair get +A / -A.  The copies carry identical activations, so the pair contributes (w + A - A) d: the
    network computes what the even channels alone would -- but interval arithmetic sees |w + A| + |A| per pair.  The load-time
    analysis then finds a comfortable bound on a tensor whose real values are ~s: the case its underflow ESTIMATE cannot decide
     and calibration on real crops does."""
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    expand, dw, proj = _block_keys(7)
    for L in (expand, dw):
        w = out[L['key'] + '.weight']
        w[1::2] = w[0::2]
        for q in ('weight', 'bias', 'running_mean', 'running_var'):
            v = out[L['bn'] + '.' + q]
            v[1::2] = v[0::2]
    wp = out[proj['key'] + '.weight']
    wp[:, 1::2] = np.float32(-A)
    wp[:, 0::2] = wp[:, 0::2] + np.float32(A)
    return scale_stream(out, '64', s)


def spread_rows(sd: dict, feature: int, decades: float = 6.0, zero_shift: bool = False, seed: int = 3) -> dict:
    """Output rows of the EXPAND convolution of .features[feature] scaled by 10^-decades .. 1 with NO compensation anywhere: the
    hidden channels really are that far apart.  zero_shift: their BatchNorm bias / running mean are zeroed too, so that a small
    row's output is its (tiny) GEMM result alone instead of its BN shift."""
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    expand, _, _ = _block_keys(feature)
    rng = np.random.default_rng(seed)
    n = out[expand['key'] + '.weight'].shape[0]
    sc = (10.0 ** rng.uniform(-decades, 0.0, n)).astype(np.float32)
    sc[rng.integers(0, n)] = 1.0
    out[expand['key'] + '.weight'] = out[expand['key'] + '.weight'] * sc[:, None, None, None]
    if zero_shift:
        out[expand['bn'] + '.bias'] = np.zeros(n, np.float32)
        out[expand['bn'] + '.running_mean'] = np.zeros(n, np.float32)
    return out


def spread_basis(pack: dict, decades: float = 3.0, seed: int = 9) -> dict:
    """3DMM pack whose shape / expression columns span 2 x decades orders of magnitude while the coefficients move the other way
    (column k x c_k, param_mean / param_std of alpha_k / c_k): the same kind of face, |alpha| from ~1e2 to ~1e8, column norms from
    1e-3 to 1e3 of the original -- the dynamic range real BFM-style packs have between u (~1e5) and PCA directions (~1e-3)."""
    rng = np.random.default_rng(seed)
    out = {k: np.array(v, copy=True) for k, v in pack.items()}
    c = (10.0 ** rng.uniform(-decades, decades, 50)).astype(np.float32)
    out['w_shp'] = out['w_shp'] * c[None, :40]
    out['w_exp'] = out['w_exp'] * c[None, 40:]
    out['param_mean'][12:62] = out['param_mean'][12:62] / c
    out['param_std'][12:62] = out['param_std'][12:62] / c
    return out


def scale_resnet_stream(sd: dict, layer: int, s: float) -> dict:
    """ResNet-50: the residual stream of `layer` (1..3) times s, exactly (power of two; ReLU commutes with positive scales):
    bn3 / downsample BN of every block of the layer x s, conv1 of blocks 1.. of the layer and conv1 / downsample conv of the next
    layer's first block / s."""
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    nblk = (3, 4, 6, 3)[layer - 1]
    f32 = np.float32
    for i in range(nblk):
        pre = f'layer{layer}.{i}'
        for p in ('weight', 'bias'):
            out[f'{pre}.bn3.{p}'] = out[f'{pre}.bn3.{p}'] * f32(s)
        if i == 0:
            for p in ('weight', 'bias'):
                out[f'{pre}.downsample.1.{p}'] = out[f'{pre}.downsample.1.{p}'] * f32(s)
        else:
            out[f'{pre}.conv1.weight'] = out[f'{pre}.conv1.weight'] * f32(1.0 / s)
    nxt = f'layer{layer + 1}.0'
    out[f'{nxt}.conv1.weight'] = out[f'{nxt}.conv1.weight'] * f32(1.0 / s)
    out[f'{nxt}.downsample.0.weight'] = out[f'{nxt}.downsample.0.weight'] * f32(1.0 / s)
    return out


def extreme_crops(B: int, seed: int = 61) -> np.ndarray:
    """uint8 crops with the degenerate images in front: all 255, all 0, checkerboard, half / half, binary noise."""
    crops = synth.make_crops(B, seed=seed)
    rng = np.random.default_rng(seed + 1)
    crops[0] = 255
    if B > 1:
        crops[1] = 0
    if B > 2:
        crops[2] = (np.indices((120, 120)).sum(0) % 2 * 255).astype(np.uint8)[:, :, None]
    if B > 3:
        crops[3, :, :60] = 255
        crops[3, :, 60:] = 0
    if B > 4:
        crops[4] = rng.integers(0, 2, (120, 120, 3), dtype=np.uint8) * 255
    return crops
