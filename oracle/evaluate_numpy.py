"""TEST INFRASTRUCTURE -- numpy restatement of the reference's AFLW2000-3D evaluator (data-gated part).
Only tests/ may import it.

  calc_nme        benchmark_aflw2000.py:107-139   (ground truth passed in: the reference loads it from aflw2000_data/eval at import)
  ana             benchmark_aflw2000.py:21-52     (yaw-binned means; returns the five numbers, prints nothing)
  foe_mae         benchmark.py:176-209            (benchmark_FOE: mean absolute Euler-angle error, [pitch, yaw, roll] order)
Pinned by tests/golden/evaluate_golden.npz, produced by the REAL reference module (its `_load` patched to serve seeded synthetic
ground truth; tests/golden/make_golden.py main_evaluate).
"""
from math import sqrt

import numpy as np


def calc_nme(pts68_fit_all, pts68_all, roi_boxs):
    std_size = 120
    nme_list = []
    for i in range(len(roi_boxs)):
        pts68_fit = np.array(pts68_fit_all[i], copy=True)
        pts68_gt = pts68_all[i]
        sx, sy, ex, ey = roi_boxs[i]
        scale_x = (ex - sx) / std_size
        scale_y = (ey - sy) / std_size
        pts68_fit[0, :] = pts68_fit[0, :] * scale_x + sx
        pts68_fit[1, :] = pts68_fit[1, :] * scale_y + sy
        minx, maxx = np.min(pts68_gt[0, :]), np.max(pts68_gt[0, :])
        miny, maxy = np.min(pts68_gt[1, :]), np.max(pts68_gt[1, :])
        llength = sqrt((maxx - minx) * (maxy - miny))
        dis = pts68_fit - pts68_gt[:2, :]
        dis = np.sqrt(np.sum(np.power(dis, 2), 0))
        nme_list.append(np.mean(dis) / llength)
    return np.array(nme_list, dtype=np.float32)


def ana(nme_list, yaws_list):
    yaw_list_abs = np.abs(yaws_list)
    i1 = yaw_list_abs <= 30
    i2 = np.bitwise_and(yaw_list_abs > 30, yaw_list_abs <= 60)
    i3 = yaw_list_abs > 60
    m = [np.mean(nme_list[i1]) * 100, np.mean(nme_list[i2]) * 100, np.mean(nme_list[i3]) * 100]
    return m[0], m[1], m[2], np.mean(m), np.std(m)


def foe_mae(angles_pyr, pose_gt):
    """benchmark.py:199-204: angles_pyr [n,3] decoded angles already swapped to [pitch, yaw, roll]; returns (MAE, yaw, pitch, roll)."""
    pa = np.mean(np.abs(angles_pyr - pose_gt), axis=0)
    return float(np.mean(pa)), float(pa[1]), float(pa[0]), float(pa[2])
