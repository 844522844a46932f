"""ORACLE (test infrastructure, NOT product code).

Host restatement of the pre-processing of get_all_outputs (reference synergy3DMM.py:187-188):
`crop_img` (utils/inference.py:95-125) and `cv2.resize(..., (120, 120), interpolation=cv2.INTER_LANCZOS4)`.
Only tests/ may import this module; it is the checker of the device kernel `syn_crop_resize`
(csrc/preproc_kernels.hip) and of the tap tables the product computes (synergynet_amd/inference.py).

PARITY UNPINNED for the resize: OpenCV is an un-vendored, unpinned dependency of the reference and is not installed
here, so `resize_lanczos4` restates OpenCV's published algorithm for 8-bit images (8-tap Lanczos a = 4 kernel evaluated through
the angle-addition table, weights normalised in float32, 11-bit fixed point, replicated borders, horizontal pass then
vertical pass on int32 sums, rounding shift by 22) and nothing here can compare it with cv2 itself.  `crop_img` is
exact by construction (integer slicing) and is pinned by the reference's own function in tests/test_oracle_golden.py
when /root/reference is present.
"""
from __future__ import annotations

import numpy as np


def crop_img(img, roi_box):
    """reference utils/inference.py:95-125: the box is rounded to integers, the part of it inside the frame is copied and
    the rest of the (ey-sy) x (ex-sx) result stays zero."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    sx, sy, ex, ey = (int(round(v)) for v in roi_box[:4])
    out = np.zeros((ey - sy, ex - sx) + img.shape[2:], dtype=np.uint8)
    x0, x1 = max(sx, 0), min(ex, w)        # source window clipped to the frame
    y0, y1 = max(sy, 0), min(ey, h)
    out[y0 - sy:y1 - sy, x0 - sx:x1 - sx] = img[y0:y1, x0:x1]
    return out


def lanczos4_taps(n_dst: int, n_src: int):
    """(first of the 8 source taps per destination index [n_dst], 11-bit fixed-point weights [n_dst,8]) -- vectorised,
    written independently of the product's per-index loop (synergynet_amd/inference.py:_lanczos4_taps).  OpenCV rounds
    the source position to float32 before taking floor and fraction, forms the kernel argument x + 3 - i in float32 and sums
    the eight weights one after the other in float32."""
    f32 = np.float32
    pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5).astype(f32)
    first = np.floor(pos).astype(np.int64)
    frac = pos - first.astype(f32)
    r = 0.70710678118654752440084436210485
    table = np.array([[1, 0], [-r, -r], [0, 1], [r, -r], [-1, 0], [r, r], [0, -1], [-r, r]])
    x3 = frac + f32(3)                                                   # float32
    ang0 = -x3.astype(np.float64) * np.pi * 0.25
    s0, c0 = np.sin(ang0), np.cos(ang0)
    arg = (x3[:, None] - np.arange(8, dtype=f32)[None, :]).astype(np.float64)        # (x + 3 - i) rounded to float32
    exact = frac < np.finfo(f32).eps
    arg[exact] = 1.0                                                     # placeholder, rows overwritten below
    y = -arg * np.pi * 0.25
    wts = ((table[None, :, 0] * s0[:, None] + table[None, :, 1] * c0[:, None]) / (y * y)).astype(f32)
    tot = np.zeros(n_dst, dtype=f32)
    for i in range(8):
        tot = (tot + wts[:, i]).astype(f32)
    wts = wts * (f32(1.0) / tot)[:, None]
    wts[exact] = 0
    wts[exact, 3] = 1
    fixed = np.clip(np.rint(wts * f32(2048.0)), -32768, 32767).astype(np.int64)
    return first - 3, fixed


def resize_lanczos4(img, out_h: int, out_w: int):
    """uint8 [h,w(,c)] -> uint8 [out_h,out_w(,c)]; see the module docstring (UNPINNED vs cv2)."""
    img = np.asarray(img)
    flat = img.ndim == 2
    src = (img[:, :, None] if flat else img).astype(np.int64)
    h, w = src.shape[:2]
    if h == 0 or w == 0:
        raise ValueError('resize_lanczos4: empty crop')
    fx, wx = lanczos4_taps(out_w, w)
    fy, wy = lanczos4_taps(out_h, h)
    cols = np.clip(fx[:, None] + np.arange(8), 0, w - 1)
    rows = np.clip(fy[:, None] + np.arange(8), 0, h - 1)
    hor = np.einsum('yxkc,xk->yxc', src[:, cols, :], wx)
    ver = np.einsum('ykxc,yk->yxc', hor[rows], wy)
    out = np.clip((ver + (1 << 21)) >> 22, 0, 255).astype(np.uint8)
    return out[:, :, 0] if flat else out
