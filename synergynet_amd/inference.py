"""Host-side pre-processing of get_all_outputs (reference synergy3DMM.py:177-192) and the
module-level helper names of the reference's utils/inference.py.

The crop (`crop_img`, utils/inference.py:95-125) and the resize (`cv2.resize(img, (120,120),
interpolation=cv2.INTER_LANCZOS4)`, synergy3DMM.py:188) run on the device (`syn_crop_resize`,
csrc/preproc_kernels.hip); the host only supplies the per-box tap tables.  `lanczos4_tables` follows
OpenCV's published 8-tap Lanczos resampler (fixed-point 8-bit path: 11-bit coefficients) from its
algorithm description: OpenCV is not installed in this environment and is an un-vendored, unpinned
dependency of the reference, so PARITY OF THESE TABLES WITH cv2 IS UNPINNED.  The host
checker of the device kernel (whole-image crop + resize in numpy) lives in oracle/preproc_numpy.py.

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


def _lanczos4_taps(n_dst: int, n_src: int):
    """Per destination index: first source tap (may be out of range, clamp later) and 8 fixed-point weights.
    OpenCV's resize evaluates the source position in double, rounds it to float32 and takes floor / fraction of THAT
    (`fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx`), so the fraction is always < 1; the kernel argument
    `x + 3 - i` is float32 arithmetic and the eight weights are summed sequentially in float32."""
    f32 = np.float32
    scale = n_src / n_dst
    pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)
    sx = np.floor(pos).astype(np.int64)
    fx = pos - sx.astype(f32)                      # exact in float32
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
    coeffs = np.zeros((n_dst, 8), dtype=f32)
    for d in range(n_dst):
        x = fx[d]
        if x < np.finfo(f32).eps:
            coeffs[d, 3] = 1.0
            continue
        x3 = f32(x + f32(3))
        y0 = -float(x3) * np.pi * 0.25
        s0, c0 = np.sin(y0), np.cos(y0)
        c = np.empty(8, dtype=f32)
        tot = f32(0)
        for i in range(8):
            y = -float(f32(x3 - f32(i))) * np.pi * 0.25
            c[i] = f32((cs[i, 0] * s0 + cs[i, 1] * c0) / (y * y))
            tot = f32(tot + c[i])
        coeffs[d] = c * f32(f32(1.0) / tot)
    icoef = np.clip(np.rint(coeffs * f32(2048.0)), -32768, 32767).astype(np.int64)
    return sx - 3, icoef


def lanczos4_tables(n_src: int, n_dst: int = 120):
    """(first tap in source coordinates [n_dst] int32, fixed-point weights [n_dst,8] int16) for syn_crop_resize."""
    x0, c = _lanczos4_taps(n_dst, n_src)
    return x0.astype(np.int32), c.astype(np.int16)


def predict_sparseVert(param, roi_box, transform=False):
    """reference utils/inference.py:140-141"""
    return _model().predict_sparseVert(param, roi_box, transform=transform)


def predict_denseVert(param, roi_box, transform=False):
    """reference utils/inference.py:143-144"""
    return _model().predict_denseVert(param, roi_box, transform=transform)


def predict_pose(param, roi_bbox, ret_mat=False):
    """reference utils/inference.py:146-157; ret_mat=True returns parse_pose's P = [R | t3d] (3,4)."""
    return _model().predict_pose(param, roi_bbox, ret_mat=ret_mat)
