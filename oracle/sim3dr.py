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


_norm = lambda arr: arr / np.sqrt(np.sum(arr ** 2, axis=1))[:, None]       # lighting.py:6


def norm_vertices(vertices):                                                 # lighting.py:9-14
    vertices -= vertices.min(0)[None, :]
    vertices /= vertices.max()
    vertices *= 2
    vertices -= vertices.max(0)[None, :] / 2
    return vertices


def convert_type(obj):                                                       # lighting.py:17-20
    if isinstance(obj, (tuple, list)):
        return np.array(obj, dtype=np.float32)[None, :]
    return obj


RENDER_CFG = dict(intensity_ambient=0.75, color_ambient=(1, 1, 1), intensity_directional=0.7, color_directional=(1, 1, 1),
                  intensity_specular=0.2, specular_exp=5, light_pos=(0, 0, 5), view_pos=(0, 0, 5))      # utils/render.py:18-27


class RenderPipeline:
    """Sim3DR/lighting.py:23-71."""

    def __init__(self, impl='oracle', **kwargs):
        self.impl = impl
        self.intensity_ambient = convert_type(kwargs.get('intensity_ambient', 0.3))
        self.intensity_directional = convert_type(kwargs.get('intensity_directional', 0.6))
        self.intensity_specular = convert_type(kwargs.get('intensity_specular', 0.1))
        self.specular_exp = kwargs.get('specular_exp', 5)
        self.color_ambient = convert_type(kwargs.get('color_ambient', (1, 1, 1)))
        self.color_directional = convert_type(kwargs.get('color_directional', (1, 1, 1)))
        self.light_pos = convert_type(kwargs.get('light_pos', (0, 0, 5)))
        self.view_pos = convert_type(kwargs.get('view_pos', (0, 0, 5)))

    def light(self, vertices, triangles):
        normal = get_normal(vertices, triangles, impl=self.impl)
        light = np.zeros_like(vertices, dtype=np.float32)
        if self.intensity_ambient > 0:
            light += self.intensity_ambient * self.color_ambient
        vertices_n = norm_vertices(vertices.copy())
        if self.intensity_directional > 0:
            direction = _norm(self.light_pos - vertices_n)
            cos = np.sum(normal * direction, axis=1)[:, None]
            light += self.intensity_directional * (self.color_directional * np.clip(cos, 0, 1))
            if self.intensity_specular > 0:
                v2v = _norm(self.view_pos - vertices_n)
                reflection = 2 * cos * normal - direction
                spe = np.sum((v2v * reflection) ** self.specular_exp, axis=1)[:, None]
                spe = np.where(cos != 0, np.clip(spe, 0, 1), np.zeros_like(spe))
                light += self.intensity_specular * self.color_directional * np.clip(spe, 0, 1)
        return np.clip(light, 0, 1)

    def __call__(self, vertices, triangles, bg, texture=None):
        light = self.light(vertices, triangles)
        if texture is None:
            return rasterize(vertices, triangles, light, bg=bg, impl=self.impl)
        texture *= light
        return rasterize(vertices, triangles, texture, bg=bg, impl=self.impl)


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
