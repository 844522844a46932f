"""Mesh consumers on the GPU: the reference's Sim3DR package and utils/render.py.

Same names, argument meaning and return types as the reference:
  get_normal(vertices, triangles)                          Sim3DR/Sim3DR.py:8-11
  rasterize(vertices, triangles, colors, bg=..., ...)      Sim3DR/Sim3DR.py:14-29
  RenderPipeline(**cfg)(vertices, triangles, bg, texture)  Sim3DR/lighting.py:23-71
  render(img, ver_lst, alpha=0.6, tex=None)                utils/render.py:31-50 (returns the blended image; no files)
numpy in, numpy out, like the reference -- but every stage runs in HIP kernels through the C ABI
(syn_load_triangles / syn_mesh_shade / syn_rasterize / syn_add_weighted, csrc/render_kernels.hip) on the most recently
constructed SynergyNet's handle; there is no CPU fallback.  `render_batch` is the device-resident entry: meshes as the
[F,3,N] tensor `reconstruct(..., dense=True)` returns, no transposes, no host copies.
"""
from __future__ import annotations

import ctypes as C
import zlib

import numpy as np
import torch

from . import abi
from . import inference as _inf

RENDER_CFG = dict(intensity_ambient=0.75, color_ambient=(1, 1, 1), intensity_directional=0.7, color_directional=(1, 1, 1),
                  intensity_specular=0.2, specular_exp=5, light_pos=(0, 0, 5), view_pos=(0, 0, 5))      # utils/render.py:18-27


def _model():
    return _inf._model()


def _triangles(m, triangles):
    """Upload the topology when it differs from the one the handle holds (the reference passes it on every call)."""
    t = np.ascontiguousarray(np.asarray(triangles), dtype=np.int32)
    if t.ndim != 2 or t.shape[1] != 3:
        raise ValueError('triangles must be [ntri,3] (0-based), as Sim3DR takes them')
    key = (t.shape[0], zlib.crc32(t.tobytes()))          # content hash: the reference passes the array on every call
    return t, key


def _ensure_topology(m, triangles, nver):
    t, key = _triangles(m, triangles)
    if getattr(m, '_tri_key', None) != (key, nver):
        abi.check(m._lib.syn_load_triangles(m._h, t.ctypes.data_as(C.c_void_p), t.shape[0], nver))
        m._tri_key = (key, nver)
    return t


def _cfg16(pipe):
    f = lambda v: [float(x) for x in np.asarray(v, dtype=np.float32).reshape(-1)]
    vals = ([float(np.float32(pipe.intensity_ambient))] + f(pipe.color_ambient) + [float(np.float32(pipe.intensity_directional))] +
            f(pipe.color_directional) + [float(np.float32(pipe.intensity_specular)), float(pipe.specular_exp)] +
            f(pipe.light_pos) + f(pipe.view_pos))
    return (C.c_float * 16)(*vals)


def _planar_arg(meshes):
    """The ABI's `planar` argument for a [F,3,N] float32 device tensor: 1 for packed rows, the row pitch for the pitched view
    reconstruct() returns ([F,3,pitch][:, :, :N]); anything else (non-unit column stride, unevenly spaced faces) is refused."""
    F, three, n = meshes.shape
    pitch = meshes.stride(1) if n > 1 else n
    if three != 3 or meshes.dtype != torch.float32 or (n > 1 and meshes.stride(2) != 1) or pitch < n or (F > 1 and meshes.stride(0) != 3 * pitch):
        raise RuntimeError('meshes must be a float32 [F,3,N] tensor with unit column stride and equally pitched rows')
    return 1 if pitch == n else int(pitch)


def _shade(m, verts_t, planar, cfg=None):
    """verts_t: device tensor [F,3,N] (planar: 1 packed, > 1 row pitch) or [F,N,3]; returns (normal, light or None) device
    tensors [F,N,3]."""
    F = verts_t.shape[0]
    n = verts_t.shape[2] if planar else verts_t.shape[1]
    normal = torch.empty((F, n, 3), dtype=torch.float32, device=m.device)
    light = torch.empty_like(normal) if cfg is not None else None
    with torch.cuda.device(m.device):
        abi.check(m._lib.syn_mesh_shade(m._h, verts_t.data_ptr(), F, int(planar), cfg, normal.data_ptr(),
                                        light.data_ptr() if light is not None else None, m._stream()))
    return normal, light


def get_normal(vertices, triangles):
    """Sim3DR/Sim3DR.py:8-11: vertices [nver,3] float32, triangles [ntri,3] int32 -> normals [nver,3] float32."""
    m = _model()
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    _ensure_topology(m, triangles, v.shape[0])
    vt = torch.from_numpy(v).to(m.device)[None]
    normal, _ = _shade(m, vt, planar=False)
    return normal[0].cpu().numpy()


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False):
    """Sim3DR/Sim3DR.py:14-29 (bg is drawn into and returned, like the reference's in-place Cython call)."""
    m = _model()
    if bg is not None:
        height, width, channel = bg.shape
    else:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)
    if bg.dtype != np.uint8:
        raise TypeError('bg must be uint8 (the reference binding takes unsigned char, rasterize.pyx:97)')
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    col = np.ascontiguousarray(colors, dtype=np.float32)
    _ensure_topology(m, triangles, v.shape[0])
    vt = torch.from_numpy(v).to(m.device)[None]
    ct = torch.from_numpy(col).to(m.device)[None]
    img = torch.from_numpy(np.ascontiguousarray(bg)).to(m.device)
    with torch.cuda.device(m.device):
        abi.check(m._lib.syn_rasterize(m._h, vt.data_ptr(), ct.data_ptr(), 1, 0, channel, img.data_ptr(), height, width,
                                       int(reverse), m._stream()))
    bg[...] = img.cpu().numpy()
    return bg


class RenderPipeline:
    """Sim3DR/lighting.py:23-71 (texture=None path; a texture multiplies the vertex colours on the host side like :69)."""

    def __init__(self, **kwargs):
        conv = lambda o: np.array(o, dtype=np.float32)[None, :] if isinstance(o, (tuple, list)) else o
        self.intensity_ambient = conv(kwargs.get('intensity_ambient', 0.3))
        self.intensity_directional = conv(kwargs.get('intensity_directional', 0.6))
        self.intensity_specular = conv(kwargs.get('intensity_specular', 0.1))
        self.specular_exp = kwargs.get('specular_exp', 5)
        self.color_ambient = conv(kwargs.get('color_ambient', (1, 1, 1)))
        self.color_directional = conv(kwargs.get('color_directional', (1, 1, 1)))
        self.light_pos = conv(kwargs.get('light_pos', (0, 0, 5)))
        self.view_pos = conv(kwargs.get('view_pos', (0, 0, 5)))

    def update_light_pos(self, light_pos):
        self.light_pos = np.array(light_pos, dtype=np.float32)[None, :]

    def light(self, vertices, triangles):
        """The vertex colours of lighting.py:40-64, [nver,3] float32."""
        m = _model()
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        _ensure_topology(m, triangles, v.shape[0])
        _, light = _shade(m, torch.from_numpy(v).to(m.device)[None], planar=False, cfg=_cfg16(self))
        return light[0].cpu().numpy()

    def __call__(self, vertices, triangles, bg, texture=None):
        light = self.light(vertices, triangles)
        if texture is None:
            return rasterize(vertices, triangles, light, bg=bg)
        texture *= light
        return rasterize(vertices, triangles, texture, bg=bg)


def render_batch(model, img, meshes, alpha=0.6, cfg=None):
    """Device-resident utils/render.py:31-50: img uint8 [H,W,3] (tensor or array), meshes [F,3,N] float32 device tensor in
    image coordinates (reconstruct(..., roi=..., dense=True)); the topology is the model's `triangles`.
    Returns (solid overlay, blended result) as uint8 device tensors."""
    pipe = RenderPipeline(**(cfg or RENDER_CFG))
    F, _, n = meshes.shape
    if getattr(model, '_tri_obj', None) is not model.triangles or getattr(model, '_tri_key', (None, None))[1] != n:
        t = np.asarray(model.triangles)                  # the class attribute is [3,ntri] (synergy3DMM.py:105), Sim3DR wants [ntri,3]
        _ensure_topology(model, np.ascontiguousarray(t.T if t.shape[0] == 3 else t), n)
        model._tri_obj = model.triangles                 # same object next time: skip the host-side comparison
    img_t = (img if isinstance(img, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img))).to(model.device)
    H, W, ch = img_t.shape
    overlap = img_t.clone()
    res = torch.empty_like(img_t)
    planar = _planar_arg(meshes)       # the pitched rows reconstruct() writes are read in place: no packed copy on the device path
    with torch.cuda.device(model.device):
        normal, light = _shade(model, meshes, planar=planar, cfg=_cfg16(pipe))
        for f0 in range(0, F, 254):                      # the z-key has an 8-bit face field; later faces overwrite earlier ones
            f1 = min(F, f0 + 254)
            abi.check(model._lib.syn_rasterize(model._h, meshes[f0:f1].data_ptr(), light[f0:f1].data_ptr(), f1 - f0, planar, ch,
                                               overlap.data_ptr(), H, W, 0, model._stream()))
        abi.check(model._lib.syn_add_weighted(model._h, img_t.data_ptr(), C.c_float(1 - alpha), overlap.data_ptr(), C.c_float(alpha),
                                              res.data_ptr(), img_t.numel(), model._stream()))
    return overlap, res


def render(img, ver_lst, alpha=0.6, wfp=None, tex=None, connectivity=None):
    """utils/render.py:31-50: img uint8 [H,W,3], ver_lst = the mesh list of get_all_outputs ((3,N) arrays).  Returns the
    blended image; `wfp` (file output through cv2.imwrite in the reference) is not supported here."""
    if wfp is not None:
        raise NotImplementedError('file output is outside this library (the reference uses cv2.imwrite)')
    if tex is not None:
        raise NotImplementedError('textured rendering: use RenderPipeline.__call__(..., texture=...)')
    m = _model()
    if connectivity is not None:
        tri = np.ascontiguousarray(np.asarray(connectivity).T, dtype=np.int32)
        meshes = torch.from_numpy(np.stack([np.asarray(v, dtype=np.float32) for v in ver_lst])).to(m.device)
        _ensure_topology(m, tri, meshes.shape[2])
        saved = m.triangles
        m.triangles = torch.from_numpy(tri.T.astype(np.int64))
        try:
            return render_batch(m, img, meshes, alpha)[1].cpu().numpy()
        finally:
            m.triangles = saved
    meshes = torch.from_numpy(np.stack([np.asarray(v, dtype=np.float32) for v in ver_lst])).to(m.device)
    return render_batch(m, img, meshes, alpha)[1].cpu().numpy()
