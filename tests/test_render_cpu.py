"""CPU tests of the mesh-consumer oracle: the C / numpy restatement of the reference's Sim3DR package
against (a) the golden fixtures produced by the REAL reference (tests/golden/make_golden.py main_render: Sim3DR/lighting.py
+ the reference's C++ rasteriser) and (b) the real reference C++ itself where /root/reference exists."""
import os

import numpy as np
import pytest

from oracle import sim3dr as osim
from synergynet_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def rgold():
    return dict(np.load(os.path.join(HERE, 'golden', 'render_golden.npz')))


def _case(cfg):
    rows, cols, nv, nt, hw, nf, seed, img_seed = [int(x) for x in cfg]
    full = rows * cols != nv
    tri = synth.make_grid_topology(rows, cols, n_vert=nv if full else None, n_tri=nt if full else None)
    meshes = synth.make_face_meshes(nf, rows, cols, n_vert=nv if full else None, height=hw, width=hw, seed=seed)
    img = np.random.default_rng(img_seed).integers(0, 256, (hw, hw, 3), dtype=np.uint8)
    return tri, meshes, img


def test_restatement_matches_reference_golden_small(rgold):
    tri, meshes, img = _case(rgold['small_cfg'])
    assert np.array_equal(img, rgold['small_img'])
    app = osim.RenderPipeline(**osim.RENDER_CFG)
    overlap = img.copy()
    for f in range(meshes.shape[0]):
        ver = np.ascontiguousarray(meshes[f].T)
        assert np.array_equal(osim.get_normal(ver, tri), rgold['small_normal'][f], equal_nan=True)      # bit-exact
        # numpy's float32 power differs in the last bit between CPU back ends (SVML / libm): 1e-6 on the vertex colours
        np.testing.assert_allclose(app.light(ver, tri), rgold['small_light'][f], rtol=0, atol=1e-6)
        overlap = app(ver, tri, overlap)
    diff = np.abs(overlap.astype(int) - rgold['small_overlay'].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-3          # one grey level on the rare pixel that crosses an integer


def test_restatement_matches_reference_golden_full_size(rgold):
    tri, meshes, img = _case(rgold['full_cfg'])
    overlap, _ = osim.render_overlay(img, [meshes[f] for f in range(meshes.shape[0])], tri)
    diff = np.abs(overlap.astype(int) - rgold['full_overlay'].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() <= 1e-3
    assert (overlap != img).any(2).mean() > 0.2                    # the two faces really cover the frame


@pytest.mark.skipif(not osim.ref_available(), reason='oracle/_ref not built (no /root/reference on this machine)')
def test_restatement_is_bit_identical_to_the_reference_cpp():
    rng = np.random.default_rng(5)
    for nv, nt, hw in ((300, 700, 64), (50, 400, 33), (4000, 9000, 128)):
        v = rng.uniform(-10, hw + 10, (nv, 3)).astype(np.float32)
        t = rng.integers(0, nv, (nt, 3)).astype(np.int32)
        t[::7, 1] = t[::7, 0]                                       # degenerate (zero-area) triangles
        t[5] = t[4]                                                 # duplicate triangle: first one wins ties
        v[::11, 2] = v[0, 2]                                        # equal depths
        a, b = osim.get_normal(v, t, 'oracle'), osim.get_normal(v, t, 'ref')
        assert np.array_equal(a, b, equal_nan=True)
        col = rng.uniform(0, 1, (nv, 3)).astype(np.float32)
        bg = rng.integers(0, 256, (hw, hw, 3), dtype=np.uint8)
        for rev in (False, True):
            ia = osim.rasterize(v, t, col, bg=bg.copy(), reverse=rev, impl='oracle')
            ib = osim.rasterize(v, t, col, bg=bg.copy(), reverse=rev, impl='ref')
            assert np.array_equal(ia, ib)


def test_isolated_vertices_get_nan_normals_like_the_reference():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5]], dtype=np.float32)
    t = np.array([[0, 1, 2]], dtype=np.int32)
    n = osim.get_normal(v, t)
    assert np.allclose(n[:3], [[0, 0, 1]] * 3) and np.isnan(n[3]).all()     # rasterize_kernel.cpp:207 leaves 0/0


def test_grid_topology_shape():
    tri = synth.make_grid_topology(n_vert=53215, n_tri=105840)
    assert tri.shape == (105840, 3) and tri.dtype == np.int32 and tri.min() == 0 and tri.max() == 53214
