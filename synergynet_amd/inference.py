"""Host-side pre-processing of get_all_outputs (reference synergy3DMM.py:177-192) and the
module-level helper names of the reference's utils/inference.py.

`crop_img` restates reference utils/inference.py:95-125.  `resize_lanczos4` stands in for
`cv2.resize(img, (120,120), interpolation=cv2.INTER_LANCZOS4)` (synergy3DMM.py:188): OpenCV is
not installed in this environment and is an un-vendored, unpinned dependency of the reference
(SURVEY 8c), so this restates OpenCV's published 8-tap Lanczos resampler (fixed-point 8-bit
path: 11-bit coefficients, replicated borders) from its algorithm description.  PARITY OF THIS
FUNCTION WITH cv2 IS UNPINNED (no cv2 here to compare against); it is outside the GPU hot path.

The vertex / pose helpers keep the reference names (`predict_sparseVert`, `predict_denseVert`,
`predict_pose`) and dispatch to the HIP kernels through the most recently constructed model.
"""
from __future__ import annotations

import numpy as np

_default_model = None


def set_default_model(model):
    global _default_model
    _default_model = model


def _model():
    if _default_model is None:
        raise RuntimeError('no SynergyNet constructed yet (the module-level helpers use the latest instance)')
    return _default_model


def crop_img(img, roi_box):
    """reference utils/inference.py:95-125: integer-rounded box, zero padding outside the image."""
    h, w = img.shape[:2]
    sx, sy, ex, ey, _ = [int(round(_)) for _ in roi_box]
    dh, dw = ey - sy, ex - sx
    if len(img.shape) == 3:
        res = np.zeros((dh, dw, 3), dtype=np.uint8)
    else:
        res = np.zeros((dh, dw), dtype=np.uint8)
    if sx < 0:
        sx, dsx = 0, -sx
    else:
        dsx = 0
    if ex > w:
        ex, dex = w, dw - (ex - w)
    else:
        dex = dw
    if sy < 0:
        sy, dsy = 0, -sy
    else:
        dsy = 0
    if ey > h:
        ey, dey = h, dh - (ey - h)
    else:
        dey = dh
    res[dsy:dey, dsx:dex] = img[sy:ey, sx:ex]
    return res


def _lanczos4_taps(n_dst: int, n_src: int):
    """Per destination index: first source tap (may be out of range, clamp later) and 8 fixed-point weights."""
    scale = n_src / n_dst
    fx = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx).astype(np.float32)
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
    coeffs = np.zeros((n_dst, 8), dtype=np.float32)
    for d in range(n_dst):
        x = float(fx[d])
        if x < np.finfo(np.float32).eps:
            coeffs[d, 3] = 1.0
            continue
        y0 = -(x + 3) * np.pi * 0.25
        s0, c0 = np.sin(y0), np.cos(y0)
        c = np.empty(8, dtype=np.float32)
        for i in range(8):
            y = -(x + 3 - i) * np.pi * 0.25
            c[i] = np.float32((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y))
        coeffs[d] = c * (np.float32(1.0) / c.sum(dtype=np.float32))
    icoef = np.clip(np.rint(coeffs * 2048.0), -32768, 32767).astype(np.int64)
    return sx - 3, icoef


def lanczos4_tables(n_src: int, n_dst: int = 120):
    """(first tap in source coordinates [n_dst] int32, fixed-point weights [n_dst,8] int16) for syn_crop_resize."""
    x0, c = _lanczos4_taps(n_dst, n_src)
    return x0.astype(np.int32), c.astype(np.int16)


def resize_lanczos4(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [h,w(,c)] -> uint8 [out_h,out_w(,c)], separable 8-tap Lanczos, replicated borders."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w = img.shape[:2]
    if h == 0 or w == 0:
        raise ValueError('resize_lanczos4: empty crop')
    x0, cx = _lanczos4_taps(out_w, w)
    y0, cy = _lanczos4_taps(out_h, h)
    src = img.astype(np.int64)
    xi = np.clip(x0[:, None] + np.arange(8)[None, :], 0, w - 1)             # [out_w,8]
    hor = (src[:, xi, :] * cx[None, :, :, None]).sum(axis=2)                # [h,out_w,c]
    yi = np.clip(y0[:, None] + np.arange(8)[None, :], 0, h - 1)             # [out_h,8]
    ver = (hor[yi] * cy[:, :, None, None]).sum(axis=1)                      # [out_h,out_w,c]
    out = np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def predict_sparseVert(param, roi_box, transform=False):
    """reference utils/inference.py:140-141"""
    return _model().predict_sparseVert(param, roi_box, transform=transform)


def predict_denseVert(param, roi_box, transform=False):
    """reference utils/inference.py:143-144"""
    return _model().predict_denseVert(param, roi_box, transform=transform)


def predict_pose(param, roi_bbox, ret_mat=False):
    """reference utils/inference.py:146-157 (ret_mat=True is not supported on the device path)."""
    if ret_mat:
        raise NotImplementedError('ret_mat=True: only (angles, t3d) is produced by the HIP pose kernel')
    return _model().predict_pose(param, roi_bbox)
