"""TEST INFRASTRUCTURE -- CPU oracle for the mesh consumers: the reference's Sim3DR package.

Only tests/, __graft_entry__.smoke() and CPU-baseline timing may import this module.

* `get_normal`, `rasterize`: the reference's Sim3DR/Sim3DR.py:8-29, dispatching to the C restatement
  oracle/sim3dr_c.c (`impl='oracle'`, built by oracle/Makefile) or to the REAL reference C++ compiled from
  /root/reference (`impl='ref'`, oracle/_ref/libsim3dr_ref.so, present only where the reference is).
* `RenderPipeline`: numpy restatement of Sim3DR/lighting.py:8-71 (same float32 numpy operations in the same order).
* `render_overlay`: utils/render.py:31-50 without the file I/O; `add_weighted` restates cv2.addWeighted's uint8 path
  (saturate(round(a*x + b*y)), OpenCV's documented formula) -- cv2 is absent here, so that last step is UNPINNED.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build():
    """Compile the C restatement (and the reference, when /root/reference exists)."""
    subprocess.run(['make', '-s', '-C', _HERE], check=True)


def _lib(impl: str):
    if impl in _libs:
        return _libs[impl]
    path = os.path.join(_HERE, '_ref/libsim3dr_ref.so' if impl == 'ref' else '_build/libsim3dr_oracle.so')
    if not os.path.isfile(path) and impl != 'ref':
        build()
    lib = C.CDLL(path)
    pre = 'ref_' if impl == 'ref' else 'sim_'
    gn, rz = getattr(lib, pre + 'get_normal'), getattr(lib, pre + 'rasterize')
    gn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    gn.restype = None
    rz.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]
    rz.restype = None
    _libs[impl] = (gn, rz)
    return _libs[impl]


def ref_available() -> bool:
    return os.path.isfile(os.path.join(_HERE, '_ref/libsim3dr_ref.so'))


def _chk(a, dtype, ndim):
    assert a.dtype == dtype and a.ndim == ndim and a.flags.c_contiguous, (a.dtype, a.shape, a.flags.c_contiguous)


def get_normal(vertices, triangles, impl='oracle'):
    """Sim3DR/Sim3DR.py:8-11.  vertices [nver,3] float32, triangles [ntri,3] int32 (C-contiguous, as the Cython binding demands)."""
    _chk(vertices, np.float32, 2); _chk(triangles, np.int32, 2)
    normal = np.zeros_like(vertices, dtype=np.float32)
    _lib(impl)[0](normal.ctypes.data, vertices.ctypes.data, triangles.ctypes.data, vertices.shape[0], triangles.shape[0])
    return normal


def rasterize(vertices, triangles, colors, bg=None, height=None, width=None, channel=None, reverse=False, impl='oracle'):
    """Sim3DR/Sim3DR.py:14-29 (alpha is left at the binding's default 1, rasterize.pyx:102)."""
    if bg is not None:
        height, width, channel = bg.shape
    else:
        assert height is not None and width is not None and channel is not None
        bg = np.zeros((height, width, channel), dtype=np.uint8)      # the binding takes unsigned char (rasterize.pyx:97)
    _chk(bg, np.uint8, 3); _chk(vertices, np.float32, 2); _chk(triangles, np.int32, 2)
    buffer = np.zeros((height, width), dtype=np.float32) - 1e8
    if colors.dtype != np.float32:
        colors = colors.astype(np.float32)
    colors = np.ascontiguousarray(colors)
    _lib(impl)[1](bg.ctypes.data, vertices.ctypes.data, triangles.ctypes.data, colors.ctypes.data, buffer.ctypes.data,
                  triangles.shape[0], height, width, channel, 1.0, int(reverse))
    return bg


RENDER_CFG = dict(intensity_ambient=0.75, color_ambient=(1, 1, 1), intensity_directional=0.7, color_directional=(1, 1, 1),
                  intensity_specular=0.2, specular_exp=5, light_pos=(0, 0, 5), view_pos=(0, 0, 5))      # utils/render.py:18-27


def _row(v):
    """lighting.py:17-20 (convert_type): tuples become float32 [1,3] rows, scalars stay python floats."""
    return np.asarray(v, dtype=np.float32)[None, :] if isinstance(v, (tuple, list)) else v


def _unit_rows(a):
    """lighting.py:6 (_norm): rows divided by their Euclidean length, in float32."""
    return a / np.sqrt(np.sum(a ** 2, axis=1))[:, None]


def vertex_colours(vertices, normal, cfg):
    """The vertex colours RenderPipeline.__call__ hands to the rasteriser (lighting.py:40-64), same float32 numpy
    operations in the same order: ambient + clipped Lambert term + (clipped) Phong specular term, on vertices normalised
    to a [-1, 1]-ish box by lighting.py:9-14."""
    k_amb, c_amb = cfg['intensity_ambient'], _row(cfg['color_ambient'])
    k_dir, c_dir = cfg['intensity_directional'], _row(cfg['color_directional'])
    k_spec, shininess = cfg['intensity_specular'], cfg['specular_exp']
    lamp, eye = _row(cfg['light_pos']), _row(cfg['view_pos'])
    colour = np.zeros_like(vertices, dtype=np.float32)
    if k_amb > 0:
        colour += k_amb * c_amb
    box = vertices.copy()                                   # norm_vertices
    box -= box.min(0)[None, :]
    box /= box.max()
    box *= 2
    box -= box.max(0)[None, :] / 2
    if k_dir > 0:
        to_lamp = _unit_rows(lamp - box)
        lambert = np.sum(normal * to_lamp, axis=1)[:, None]
        colour += k_dir * (c_dir * np.clip(lambert, 0, 1))
        if k_spec > 0:
            to_eye = _unit_rows(eye - box)
            mirrored = 2 * lambert * normal - to_lamp
            gloss = np.sum((to_eye * mirrored) ** shininess, axis=1)[:, None]
            gloss = np.where(lambert != 0, np.clip(gloss, 0, 1), np.zeros_like(gloss))
            colour += k_spec * c_dir * np.clip(gloss, 0, 1)
    return np.clip(colour, 0, 1)


class RenderPipeline:
    """Sim3DR/lighting.py:23-71 (defaults of :25-32)."""
    DEFAULTS = dict(intensity_ambient=0.3, intensity_directional=0.6, intensity_specular=0.1, specular_exp=5,
                    color_ambient=(1, 1, 1), color_directional=(1, 1, 1), light_pos=(0, 0, 5), view_pos=(0, 0, 5))

    def __init__(self, impl='oracle', **kwargs):
        self.impl = impl
        self.cfg = {k: kwargs.get(k, d) for k, d in self.DEFAULTS.items()}

    def light(self, vertices, triangles):
        return vertex_colours(vertices, get_normal(vertices, triangles, impl=self.impl), self.cfg)

    def __call__(self, vertices, triangles, bg, texture=None):
        colours = self.light(vertices, triangles)
        if texture is not None:
            texture *= colours
            colours = texture
        return rasterize(vertices, triangles, colours, bg=bg, impl=self.impl)


def add_weighted(a, alpha, b, beta):
    """cv2.addWeighted(a, alpha, b, beta, 0) for uint8 images: saturate_cast<uchar>(a*alpha + b*beta), rounding to nearest
    even as cvRound does.  OpenCV evaluates the uint8 case in float32."""
    v = a.astype(np.float32) * np.float32(alpha) + b.astype(np.float32) * np.float32(beta)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def render_overlay(img, ver_lst, tri, alpha=0.6, impl='oracle'):
    """utils/render.py:31-50 without imwrite: ver_lst = list of (3,N) float arrays (get_all_outputs meshes), tri [ntri,3] int32
    0-based.  Returns (solid overlay, alpha-blended result)."""
    app = RenderPipeline(impl=impl, **RENDER_CFG)
    overlap = img.copy()
    for ver_ in ver_lst:
        ver = np.ascontiguousarray(ver_.astype(np.float32).T)
        overlap = app(ver, tri, overlap)
    return overlap, add_weighted(img, 1 - alpha, overlap, alpha)
