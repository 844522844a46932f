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

import functools

import numpy as np

_default_model = None


def set_default_model(model):
    global _default_model
    _default_model = model


def _model():
    if _default_model is None:
        raise RuntimeError('no SynergyNet constructed yet (the module-level helpers use the latest instance)')
    return _default_model


def write_obj(obj_name, vertices, triangles):
    """Wavefront OBJ export of one mesh (reference utils/inference.py:8-23): `vertices` [3, N] as lines
    `v x y z` with four decimals, `triangles` [3, M] (1-based, as the reference's tri.mat holds them) as `f i2 i1 i0` -- the
    reference writes the corners in reverse order -- and `.obj` appended to a name that lacks it.  Host code like the
    reference's (the mesh comes down with get_all_outputs anyway); one formatted block per array instead of a Python loop
    per line: byte-identical files (tests/test_host_cpu.py against the reference's own output), ~30x faster on a 53215-vertex
    mesh."""
    if obj_name.split('.')[-1] != 'obj':
        obj_name = obj_name + '.obj'
    v = np.asarray(vertices)
    t = np.asarray(triangles)
    with open(obj_name, 'w') as f:
        if v.shape[1]:
            # '%.4f' % x formats the double value of x exactly like '{:.4f}'.format(x) does for numpy and Python floats
            f.write(('v %.4f %.4f %.4f\n' * v.shape[1]) % tuple(v[:3].T.astype(np.float64).ravel().tolist()))
        if t.shape[1]:
            rev = t[[2, 1, 0]].T.ravel()
            if np.issubdtype(t.dtype, np.integer):
                f.write(('f %d %d %d\n' * t.shape[1]) % tuple(rev.tolist()))
            else:                                   # '{}'.format of a float index prints its repr ('12.0'): kept
                f.write(('f {} {} {}\n' * t.shape[1]).format(*rev))


def _lanczos4_taps(n_dst: int, n_src: int):
    """Per destination index: first source tap (may be out of range, clamp later) and 8 fixed-point weights.
    OpenCV's resize evaluates the source position in double, rounds it to float32 and takes floor / fraction of THAT
    (`fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx`), so the fraction is always < 1; the kernel argument
    `x + 3 - i` is float32 arithmetic and the eight weights are summed sequentially in float32.  All n_dst indices at once (numpy);
    the rounding points are exactly those of the scalar formulation: float32 where OpenCV holds a float, float64 in between."""
    f32 = np.float32
    scale = n_src / n_dst
    pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(f32)
    sx = np.floor(pos).astype(np.int64)
    fx = pos - sx.astype(f32)                      # exact in float32
    s45 = 0.70710678118654752440084436210485
    cs = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
    on_grid = fx < np.finfo(f32).eps               # the source position IS a sample: weight 1 on tap 3
    x3 = (fx + f32(3)).astype(f32)
    y0 = -x3.astype(np.float64) * np.pi * 0.25
    s0, c0 = np.sin(y0), np.cos(y0)
    arg = (x3[:, None] - np.arange(8, dtype=f32)[None, :]).astype(f32).astype(np.float64)       # f32(x3 - i)
    arg[on_grid] = 1.0                             # (never used: those rows are overwritten below; avoids 0 / 0)
    y = -arg * np.pi * 0.25
    c = ((cs[None, :, 0] * s0[:, None] + cs[None, :, 1] * c0[:, None]) / (y * y)).astype(f32)
    tot = np.zeros(n_dst, dtype=f32)
    for i in range(8):                             # sequential float32 sum, tap 0 first
        tot = (tot + c[:, i]).astype(f32)
    coeffs = (c * (f32(1.0) / tot)[:, None]).astype(f32)
    coeffs[on_grid] = 0
    coeffs[on_grid, 3] = 1
    icoef = np.clip(np.rint(coeffs * f32(2048.0)), -32768, 32767).astype(np.int64)
    return sx - 3, icoef


@functools.lru_cache(maxsize=4096)
def _lanczos4_tables_cached(n_src: int, n_dst: int):
    x0, c = _lanczos4_taps(n_dst, n_src)
    x0, c = x0.astype(np.int32), c.astype(np.int16)
    x0.setflags(write=False)
    c.setflags(write=False)
    return x0, c


def lanczos4_tables(n_src: int, n_dst: int = 120):
    """(first tap in source coordinates [n_dst] int32, fixed-point weights [n_dst,8] int16) for syn_crop_resize.  Depends on the
    crop side only, so the tables are computed once per size and cached (read-only arrays): a face costs two dictionary look-ups
    instead of 4 ms of per-index Python (the reference resizes with cv2 at synergy3DMM.py:188)."""
    return _lanczos4_tables_cached(int(n_src), int(n_dst))


def crop_img(img, roi_box):
    """reference utils/inference.py:95-125 (imported from there by the reference's demo scripts): the box is rounded to integers,
    the part of it inside the frame is copied, the rest of the (ey - sy) x (ex - sx) result stays zero.  Host helper kept for
    scripts ported by module mapping; get_all_outputs crops on the device (syn_crop_resize).  A box entirely outside the frame
    gives an all-zero crop (the reference's slice assignment raises there)."""
    img = np.asarray(img)
    h, w = img.shape[:2]
    sx, sy, ex, ey = (int(round(v)) for v in roi_box[:4])
    res = np.zeros((ey - sy, ex - sx) + img.shape[2:], dtype=np.uint8)
    dsx, dsy = max(0, -sx), max(0, -sy)            # where the part inside the frame lands in the result
    sx, sy, ex, ey = max(sx, 0), max(sy, 0), min(ex, w), min(ey, h)
    if ex > sx and ey > sy:
        res[dsy:dsy + ey - sy, dsx:dsx + ex - sx] = img[sy:ey, sx:ex]
    return res


def predict_sparseVert(param, roi_box, transform=False):
    """reference utils/inference.py:140-141"""
    return _model().predict_sparseVert(param, roi_box, transform=transform)


def predict_denseVert(param, roi_box, transform=False):
    """reference utils/inference.py:143-144"""
    return _model().predict_denseVert(param, roi_box, transform=transform)


def predict_pose(param, roi_bbox, ret_mat=False):
    """reference utils/inference.py:146-157; ret_mat=True returns parse_pose's P = [R | t3d] (3,4)."""
    return _model().predict_pose(param, roi_bbox, ret_mat=ret_mat)
