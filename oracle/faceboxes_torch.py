"""TEST INFRASTRUCTURE -- CPU oracle of the FaceBoxes detector.  Only tests/ and smoke() may import it.

torch.nn.functional / numpy restatement, function by function, of the reference's detector:
  net_forward      FaceBoxes/models/faceboxes.py:116-150 (FaceBoxesNet.forward, phase='test'), :8-18 BasicConv2d,
                   :21-46 Inception, :48-61 CRelu
  prior_boxes      FaceBoxes/utils/prior_box.py:10-48
  decode           FaceBoxes/utils/box_utils.py:177-196
  cpu_nms          FaceBoxes/utils/nms/cpu_nms.pyx:17-68 (the variant nms_wrapper.py:14-19 dispatches to: suppress on ovr >= thresh)
  detect           FaceBoxes/FaceBoxes.py:60-143 (FaceBoxes.__call__), including the down-scaling of large frames;
                   cv2.resize (bilinear, uint8) is restated from OpenCV's fixed-point algorithm in `resize_linear_u8`
                   (cv2 is absent here: that step is UNPINNED).
Pinned by tests/golden/faceboxes_golden.npz, produced by the REAL reference modules (tests/golden/make_golden.py
main_faceboxes) on seeded weights and frames.
"""
from __future__ import annotations

from math import ceil

import numpy as np
import torch
import torch.nn.functional as F

CFG = dict(min_sizes=[[32, 64, 128], [256], [512]], steps=[32, 64, 128], variance=[0.1, 0.2], clip=False)     # utils/config.py:3-9
CONFIDENCE_THRESHOLD, TOP_K, KEEP_TOP_K, NMS_THRESHOLD, VIS_THRES = 0.05, 5000, 750, 0.3, 0.5                 # FaceBoxes.py:18-22
HEIGHT, WIDTH = 720, 1080                                                                                    # FaceBoxes.py:26


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _basic(sd, p, x, stride=1, padding=0):            # faceboxes.py:8-18
    x = F.conv2d(x, _t(sd, p + '.conv.weight'), None, stride, padding)
    x = F.batch_norm(x, _t(sd, p + '.bn.running_mean'), _t(sd, p + '.bn.running_var'), _t(sd, p + '.bn.weight'), _t(sd, p + '.bn.bias'),
                     False, 0.0, 1e-5)
    return F.relu(x)


def _crelu(sd, p, x, stride, padding):                # faceboxes.py:48-61
    x = F.conv2d(x, _t(sd, p + '.conv.weight'), None, stride, padding)
    x = F.batch_norm(x, _t(sd, p + '.bn.running_mean'), _t(sd, p + '.bn.running_var'), _t(sd, p + '.bn.weight'), _t(sd, p + '.bn.bias'),
                     False, 0.0, 1e-5)
    return F.relu(torch.cat([x, -x], 1))


def _inception(sd, p, x):                             # faceboxes.py:21-46
    b1 = _basic(sd, p + 'branch1x1', x)
    b2 = _basic(sd, p + 'branch1x1_2', F.avg_pool2d(x, kernel_size=3, stride=1, padding=1))
    b3 = _basic(sd, p + 'branch3x3', _basic(sd, p + 'branch3x3_reduce', x), padding=1)
    b4 = _basic(sd, p + 'branch3x3_3', _basic(sd, p + 'branch3x3_2', _basic(sd, p + 'branch3x3_reduce_2', x), padding=1), padding=1)
    return torch.cat([b1, b2, b3, b4], 1)


def net_forward(sd, x):
    """x [1,3,H,W] float32 (BGR minus mean) -> (loc [1,P,4], conf [1,P,2] softmax)."""
    with torch.no_grad():
        src = []
        x = _crelu(sd, 'conv1', x, 4, 3)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        x = _crelu(sd, 'conv2', x, 2, 2)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for i in (1, 2, 3):
            x = _inception(sd, f'inception{i}.', x)
        src.append(x)
        x = _basic(sd, 'conv3_2', _basic(sd, 'conv3_1', x), stride=2, padding=1)
        src.append(x)
        x = _basic(sd, 'conv4_2', _basic(sd, 'conv4_1', x), stride=2, padding=1)
        src.append(x)
        loc, conf = [], []
        for i, s in enumerate(src):
            loc.append(F.conv2d(s, _t(sd, f'loc.{i}.weight'), _t(sd, f'loc.{i}.bias'), 1, 1).permute(0, 2, 3, 1).contiguous())
            conf.append(F.conv2d(s, _t(sd, f'conf.{i}.weight'), _t(sd, f'conf.{i}.bias'), 1, 1).permute(0, 2, 3, 1).contiguous())
        loc = torch.cat([o.view(o.size(0), -1) for o in loc], 1)
        conf = torch.cat([o.view(o.size(0), -1) for o in conf], 1)
        return loc.view(loc.size(0), -1, 4), torch.softmax(conf.view(conf.size(0), -1, 2), dim=-1)


def prior_boxes(image_size):
    """prior_box.py:10-48: [P,4] float32 (cx, cy, w, h) in image-relative units.  The reference builds the list with python
    floats cell by cell; the same double-precision expressions evaluated array-wise (then cast to float32, as torch.Tensor
    does) give bit-identical rows in the same order: per cell 16 anchors of 32 px (4x4 dense, y-major), 4 of 64 px (2x2),
    1 of 128 px on the stride-32 map, then one 256 px anchor per stride-64 cell and one 512 px anchor per stride-128 cell."""
    H, W = float(image_size[0]), float(image_size[1])
    rows = []
    for step, sizes in zip(CFG['steps'], CFG['min_sizes']):
        fh, fw = ceil(image_size[0] / step), ceil(image_size[1] / step)
        ii, jj = np.meshgrid(np.arange(fh, dtype=np.float64), np.arange(fw, dtype=np.float64), indexing='ij')
        per_cell = []
        for size in sizes:
            dens = {32: 4, 64: 2}.get(size, 0)
            offs = [k / dens for k in range(dens)] if dens else [0.5]
            for oy in offs:
                for ox in offs:
                    cx = (jj + ox) * step / W
                    cy = (ii + oy) * step / H
                    per_cell.append(np.stack([cx, cy, np.full_like(cx, size / W), np.full_like(cy, size / H)], -1))
        rows.append(np.stack(per_cell, 2).reshape(-1, 4))            # [fh, fw, anchors, 4] -> cell-major, anchor-minor
    return torch.from_numpy(np.concatenate(rows, 0).astype(np.float32))


def decode(loc, priors, variances):                   # box_utils.py:177-196
    boxes = torch.cat((priors[:, :2] + loc[:, :2] * variances[0] * priors[:, 2:],
                       priors[:, 2:] * torch.exp(loc[:, 2:] * variances[1])), 1)
    boxes[:, :2] -= boxes[:, 2:] / 2
    boxes[:, 2:] += boxes[:, :2]
    return boxes


def cpu_nms(dets, thresh):                            # cpu_nms.pyx:17-68 (float32 arithmetic)
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    n = dets.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0.0), xx2 - xx1 + 1)
        h = np.maximum(np.float32(0.0), yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return keep


def resize_linear_u8(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) for uint8 (INTER_LINEAR, the default): OpenCV's fixed-point path -- 11-bit coefficients,
    horizontal pass in int32, vertical pass ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2.  UNPINNED (cv2 absent)."""
    h, w = img.shape[:2]

    def taps(n_dst, n_src):
        scale = n_src / n_dst
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)     # fx = (float)((dx+0.5)*scale_x - 0.5)
        s = np.floor(f).astype(np.int64)                                                      # sx = cvFloor(fx)
        f = f - s.astype(np.float32)                                                          # fx -= sx
        lo = s < 0
        f[lo] = 0.0; s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0.0; s[hi] = n_src - 1
        c1 = np.rint(f * 2048.0).astype(np.int64)             # saturate_cast<short>(f * INTER_RESIZE_COEF_SCALE)
        c0 = np.rint((1.0 - f) * 2048.0).astype(np.int64)
        return s, np.minimum(s + 1, n_src - 1), c0, c1

    x0, x1, cx0, cx1 = taps(out_w, w)
    y0, y1, cy0, cy1 = taps(out_h, h)
    src = img.astype(np.int64)
    hor = src[:, x0, :] * cx0[None, :, None] + src[:, x1, :] * cx1[None, :, None]          # [h,out_w,c]
    s0, s1 = hor[y0], hor[y1]
    out = (((cy0[:, None, None] * (s0 >> 4)) >> 16) + ((cy1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def detect(sd, img_, return_all=False):
    """FaceBoxes.__call__ (FaceBoxes.py:60-143): uint8 BGR frame -> [[xmin, ymin, xmax, ymax, score], ...] with score > 0.5
    (return_all: the rows before the vis_thres filter as a float32 array)."""
    img_raw = img_.copy()
    scale = 1
    h, w = img_raw.shape[:2]
    if h > HEIGHT:
        scale = HEIGHT / h
    if w * scale > WIDTH:
        scale *= WIDTH / (w * scale)
    img_raw_scale = img_raw if scale == 1 else resize_linear_u8(img_raw, int(scale * h), int(scale * w))
    img = np.float32(img_raw_scale)
    im_height, im_width, _ = img.shape
    scale_bbox = torch.Tensor([img.shape[1], img.shape[0], img.shape[1], img.shape[0]])
    img -= (104, 117, 123)
    x = torch.from_numpy(img.transpose(2, 0, 1)).unsqueeze(0)
    loc, conf = net_forward(sd, x)
    priors = prior_boxes((im_height, im_width))
    boxes = decode(loc.squeeze(0), priors, CFG['variance'])
    boxes = boxes * scale_bbox / scale / 1
    boxes = boxes.numpy()
    scores = conf.squeeze(0).numpy()[:, 1]
    inds = np.where(scores > CONFIDENCE_THRESHOLD)[0]
    boxes, scores = boxes[inds], scores[inds]
    order = scores.argsort()[::-1][:TOP_K]
    boxes, scores = boxes[order], scores[order]
    dets = np.hstack((boxes, scores[:, np.newaxis])).astype(np.float32, copy=False)
    keep = cpu_nms(dets, NMS_THRESHOLD) if dets.shape[0] else []
    dets = dets[keep, :][:KEEP_TOP_K, :]
    if return_all:
        return dets
    return [[b[0], b[1], b[2], b[3], b[4]] for b in dets if b[4] > VIS_THRES]
