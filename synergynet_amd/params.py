"""3DMM constants pack: host-side mirror of the reference's utils/params.py:ParamsPack.

Same attribute names (keypoints, w_shp, w_exp, param_mean, param_std, u_shp, u_exp, u, w,
w_base, w_norm, w_base_norm, u_base, w_shp_base, w_exp_base, std_size, dim;
reference utils/params.py:13-35) and the same 'Missing data' error (:36-37).  The pack
can also be built from an in-memory dict (synthetic assets: the real 3dmm_data/ files are
Google-Drive downloads that are absent from the reference checkout).
"""
from __future__ import annotations

import os
import pickle

import numpy as np

_FILES = ('keypoints_sim.npy', 'w_shp_sim.npy', 'w_exp_sim.npy', 'param_whitening.pkl', 'u_shp.npy', 'u_exp.npy')


def _load(fp):
    """reference utils/io.py:22-27"""
    if fp.endswith('.npy'):
        return np.load(fp)
    if fp.endswith('.pkl'):
        with open(fp, 'rb') as f:
            return pickle.load(f)
    raise ValueError(fp)


def default_data_dir() -> str:
    """<repo>/3dmm_data, next to the package like the reference's prefix_path (synergy3DMM.py:27,73)."""
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '3dmm_data')


class ParamsPack:
    def __init__(self, data_dir: str | None = None, pack: dict | None = None):
        try:
            if pack is None:
                d = data_dir or default_data_dir()
                meta = _load(os.path.join(d, 'param_whitening.pkl'))
                pack = dict(keypoints=_load(os.path.join(d, 'keypoints_sim.npy')),
                            w_shp=_load(os.path.join(d, 'w_shp_sim.npy')), w_exp=_load(os.path.join(d, 'w_exp_sim.npy')),
                            param_mean=meta.get('param_mean'), param_std=meta.get('param_std'),
                            u_shp=_load(os.path.join(d, 'u_shp.npy')), u_exp=_load(os.path.join(d, 'u_exp.npy')))
                tri_fp = os.path.join(d, 'tri.mat')
                if os.path.isfile(tri_fp):
                    import scipy.io as sio
                    pack['tri'] = sio.loadmat(tri_fp)['tri']
            self.keypoints = np.asarray(pack['keypoints'])
            self.w_shp = np.asarray(pack['w_shp'])
            self.w_exp = np.asarray(pack['w_exp'])
            self.param_mean = np.asarray(pack['param_mean'])
            self.param_std = np.asarray(pack['param_std'])
            self.u_shp = np.asarray(pack['u_shp'])
            self.u_exp = np.asarray(pack['u_exp'])
            self.u = self.u_shp + self.u_exp                                   # params.py:25
            self.w = np.concatenate((self.w_shp, self.w_exp), axis=1)          # params.py:26
            self.w_base = self.w[self.keypoints]
            self.w_norm = np.linalg.norm(self.w, axis=0)
            self.w_base_norm = np.linalg.norm(self.w_base, axis=0)
            self.u_base = self.u[self.keypoints].reshape(-1, 1)
            self.w_shp_base = self.w_shp[self.keypoints]
            self.w_exp_base = self.w_exp[self.keypoints]
            self.std_size = 120
            self.dim = self.w_shp.shape[0] // 3
            self.tri = pack.get('tri')                                         # 1-based [3,n_tri] (synergy3DMM.py:73)
        except Exception:
            raise RuntimeError('Missing data')
