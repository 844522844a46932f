"""AFLW2000-3D evaluation on the device (data-gated): the reference's benchmark_aflw2000.py / benchmark.py.

The reference loads its ground truth from `aflw2000_data/eval/*.npy` at import time; those files (and trained weights) are not
available here, so the ground truth is passed in.  Names and arithmetic follow the reference:
  calc_nme(model, pts68_fit_all, pts68_all, roi_boxs)      benchmark_aflw2000.py:107-139 (HIP kernel through syn_nme)
  ana(nme_list, yaws_list)                                 benchmark_aflw2000.py:21-52 (five numbers; no printing)
  benchmark_aflw2000_params(model, params, pts68_all, roi_boxs, yaws_list)   benchmark.py:145-174
  benchmark_FOE(model, params, pose_GT, skip_indices)      benchmark.py:177-209
"""
from __future__ import annotations

import numpy as np
import torch

from . import abi


def calc_nme(model, pts68_fit_all, pts68_all, roi_boxs):
    fit = torch.as_tensor(np.asarray(pts68_fit_all, dtype=np.float32)[:, :2, :]).contiguous().to(model.device)
    gt = torch.as_tensor(np.asarray(pts68_all, dtype=np.float32)).contiguous().to(model.device)
    roi = torch.as_tensor(np.asarray(roi_boxs, dtype=np.float32)[:, :4]).contiguous().to(model.device)
    n = fit.shape[0]
    if gt.shape != (n, 3, 68) or fit.shape != (n, 2, 68) or roi.shape != (n, 4):
        raise ValueError('expected fit [N,2,68], ground truth [N,3,68], roi [N,4]')
    out = torch.empty((n,), dtype=torch.float32, device=model.device)
    with torch.cuda.device(model.device):
        abi.check(model._lib.syn_nme(model._h, fit.data_ptr(), gt.data_ptr(), roi.data_ptr(), out.data_ptr(), n, model._stream()))
    return out.cpu().numpy()


def ana(nme_list, yaws_list):
    yaw_list_abs = np.abs(yaws_list)
    i1 = yaw_list_abs <= 30
    i2 = np.bitwise_and(yaw_list_abs > 30, yaw_list_abs <= 60)
    i3 = yaw_list_abs > 60
    m = [np.mean(nme_list[i1]) * 100, np.mean(nme_list[i2]) * 100, np.mean(nme_list[i3]) * 100]
    return m[0], m[1], m[2], np.mean(m), np.std(m)


def benchmark_aflw2000_params(model, params, pts68_all, roi_boxs, yaws_list):
    """benchmark.py:145-174: whitened params [N,62] -> 68 landmarks in crop coordinates (one batched launch) -> NME statistics."""
    lm = model.reconstruct(torch.as_tensor(np.asarray(params, dtype=np.float32)), roi=None, dense=False, transform=True)
    return ana(calc_nme(model, lm[:, :2, :].cpu().numpy(), pts68_all, roi_boxs), yaws_list)


def benchmark_FOE(model, params, pose_GT, skip_indices):
    """benchmark.py:177-209: mean absolute Euler-angle error; returns (MAE, yaw, pitch, roll) in degrees."""
    params = np.asarray(params, dtype=np.float32)
    keep = np.array([i for i in range(params.shape[0]) if i not in set(int(s) for s in skip_indices)], dtype=np.int64)
    ang, _ = model.predict_pose_batch(torch.from_numpy(params[keep]))
    ang = ang.cpu().numpy()
    pyr = np.stack([ang[:, 1], ang[:, 0], ang[:, 2]], 1)          # :197 "we decode raw-pitch-yaw order"
    pa = np.mean(np.abs(pyr - np.asarray(pose_GT)), axis=0)
    return float(np.mean(pa)), float(pa[1]), float(pa[0]), float(pa[2])
