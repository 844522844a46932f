"""ORACLE (test infrastructure, NOT product code).

CPU/numpy restatement of the reference's 3DMM parameter -> geometry path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (synergynet_amd/) never does.

Pinned against the real reference code by tests/golden/make_golden.py (run in
the authoring container where /root/reference exists) -> tests/golden/*.npz,
checked by tests/test_oracle_golden.py.

Each function cites the reference lines it restates.
"""
from __future__ import annotations

from math import asin, atan2, cos

import numpy as np

STD_SIZE = 120  # reference utils/params.py:34


class Basis:
    """Derived 3DMM constants (reference utils/params.py:25-35)."""

    def __init__(self, pack: dict):
        self.keypoints = np.asarray(pack['keypoints']).astype(np.int64)
        self.w_shp = np.asarray(pack['w_shp'], dtype=np.float32)
        self.w_exp = np.asarray(pack['w_exp'], dtype=np.float32)
        self.param_mean = np.asarray(pack['param_mean'], dtype=np.float32)
        self.param_std = np.asarray(pack['param_std'], dtype=np.float32)
        self.u = (np.asarray(pack['u_shp']) + np.asarray(pack['u_exp'])).astype(np.float32)   # params.py:25
        self.u_base = self.u[self.keypoints].reshape(-1, 1)                                   # params.py:31
        self.w_shp_base = self.w_shp[self.keypoints]                                          # params.py:32
        self.w_exp_base = self.w_exp[self.keypoints]                                          # params.py:33
        self.std_size = STD_SIZE
        self.dim = self.w_shp.shape[0] // 3


def parse_param(param):
    """reference utils/inference.py:25-31 (62-vector -> p[3,3], offset[3,1], a_shp[40,1], a_exp[10,1])."""
    p_ = param[:12].reshape(3, 4)
    p = p_[:, :3]
    offset = p_[:, -1].reshape(3, 1)
    alpha_shp = param[12:52].reshape(40, 1)
    alpha_exp = param[52:62].reshape(10, 1)
    return p, offset, alpha_shp, alpha_exp


def param2vert(b: Basis, param, dense=False, transform=True):
    """reference utils/inference.py:64-84 (single face, numpy)."""
    if param.shape[0] != 62:
        raise RuntimeError('length of params mismatch')
    param_ = param * b.param_std[:62] + b.param_mean[:62]
    p, offset, alpha_shp, alpha_exp = parse_param(param_)
    if dense:
        s = b.u + b.w_shp @ alpha_shp + b.w_exp @ alpha_exp
    else:
        s = b.u_base + b.w_shp_base @ alpha_shp + b.w_exp_base @ alpha_exp
    vertex = p @ s.reshape(3, -1, order='F') + offset
    if transform:
        vertex[1, :] = b.std_size + 1 - vertex[1, :]
    return vertex


def predict_vertices(b: Basis, param, roi_bbox, dense, transform=True):
    """reference utils/inference.py:127-138 (ROI affine back to image coordinates)."""
    vertex = param2vert(b, param, dense=dense, transform=transform)
    sx, sy, ex, ey, _ = roi_bbox
    scale_x = (ex - sx) / 120
    scale_y = (ey - sy) / 120
    vertex[0, :] = vertex[0, :] * scale_x + sx
    vertex[1, :] = vertex[1, :] * scale_y + sy
    s = (scale_x + scale_y) / 2
    vertex[2, :] *= s
    return vertex


def P2sRt(P):
    """reference utils/inference.py:33-43."""
    t3d = P[:, 3]
    R1 = P[0:1, :3]
    R2 = P[1:2, :3]
    s = (np.linalg.norm(R1) + np.linalg.norm(R2)) / 2.0
    r1 = R1 / np.linalg.norm(R1)
    r2 = R2 / np.linalg.norm(R2)
    r3 = np.cross(r1, r2)
    R = np.concatenate((r1, r2, r3), 0)
    return s, R, t3d


def matrix2angle_corr(R):
    """reference utils/inference.py:45-62 (Euler angles in degrees, gimbal branch kept)."""
    if R[2, 0] != 1 and R[2, 0] != -1:
        x = asin(R[2, 0])
        y = atan2(R[1, 2] / cos(x), R[2, 2] / cos(x))
        z = atan2(R[0, 1] / cos(x), R[0, 0] / cos(x))
    else:
        z = 0
        if R[2, 0] == -1:
            x = np.pi / 2
            y = z + atan2(R[0, 1], R[0, 2])
        else:
            x = -np.pi / 2
            y = -z + atan2(-R[0, 1], -R[0, 2])
    return [x * 180 / np.pi, y * 180 / np.pi, z * 180 / np.pi]


def predict_pose(b: Basis, param, roi_bbox):
    """reference utils/inference.py:86-92,146-157 -> ([rx,ry,rz] degrees, t3d[3])."""
    param = param * b.param_std[:62] + b.param_mean[:62]
    Ps = param[:12].reshape(3, -1)
    _, R, t3d = P2sRt(Ps)
    pose = matrix2angle_corr(R)
    sx, sy, ex, ey, _ = roi_bbox
    scale_x = (ex - sx) / 120
    scale_y = (ey - sy) / 120
    t3d = t3d.copy()
    t3d[0] = t3d[0] * scale_x + sx
    t3d[1] = t3d[1] * scale_y + sy
    return pose, t3d


def pose_matrix(b: Basis, param):
    """reference utils/inference.py:86-92 + :155-156: predict_pose(..., ret_mat=True) returns parse_pose's
    P = concatenate(R, t3d) -- a copy made BEFORE predict_pose applies the ROI affine to t3d, so the translation column is
    the de-whitened one."""
    param = param * b.param_std[:62] + b.param_mean[:62]
    Ps = param[:12].reshape(3, -1)
    _, R, t3d = P2sRt(Ps)
    return np.concatenate((R, t3d.reshape(3, -1)), axis=1)


def reconstruct_vertex_62(b: Basis, param, dense=False, transform=True):
    """reference synergy3DMM.py:116-149 (batched; [B,62] -> [B,3,68|N]); no ROI affine.

    Restated with one GEMM over the batch instead of torch's broadcast-batched
    matmul; float32 throughout like the reference buffers.
    """
    param = np.asarray(param, dtype=np.float32)
    if param.shape[1] != 62:
        raise RuntimeError('length of params mismatch')
    param_ = param * b.param_std[:62] + b.param_mean[:62]
    p_ = param_[:, :12].reshape(-1, 3, 4)                 # synergy3DMM.py:30-37
    p, offset = p_[:, :, :3], p_[:, :, 3:4]
    alpha = param_[:, 12:62]
    if dense:
        w = np.concatenate([b.w_shp, b.w_exp], axis=1)
        u = b.u
    else:
        w = np.concatenate([b.w_shp_base, b.w_exp_base], axis=1)
        u = b.u_base
    s = alpha @ w.T + u.reshape(1, -1)                    # [B,3n]
    v = s.reshape(param.shape[0], -1, 3).transpose(0, 2, 1)   # .view(-1,n,3).transpose(1,2)
    vertex = p @ v + offset
    if transform:
        vertex[:, 1, :] = b.std_size + 1 - vertex[:, 1, :]
    return vertex.astype(np.float32)
