"""ORACLE (test infrastructure, NOT product code).

CPU restatement of the reference MobileNetV2 backbone forward
(reference backbone_nets/mobilenetv2_backbone.py:33-74 ConvBNReLU /
InvertedResidual, :107-117 cfg, :147-158 heads, :173-189 _forward_impl) written
with torch.nn.functional fp32 CPU ops from a plain state_dict, i.e. the same
third-party arithmetic (PyTorch conv2d / batch_norm / hardtanh / linear) the
reference runs on its CPU path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Pinned against the real reference module by
tests/golden/make_golden.py -> tests/golden/*.npz.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from synergynet_amd.synth import mbv2_layers, HEADS

BN_EPS = 1e-5   # nn.BatchNorm2d default used at mobilenetv2_backbone.py:37-41


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def _conv_bn(sd, L, x):
    w = _t(sd, L['key'] + '.weight')
    if L['kind'] == 'dw':
        x = F.conv2d(x, w, None, L['stride'], 1, 1, L['cout'])          # groups=hidden_dim (:62)
    elif L['kind'] == 'stem':
        x = F.conv2d(x, w, None, L['stride'], 1)                         # 3x3 s2 pad 1 (:129)
    else:
        x = F.conv2d(x, w, None, 1, 0)                                   # 1x1 (:60,:64,:140)
    bn = L['bn']
    x = F.batch_norm(x, _t(sd, bn + '.running_mean'), _t(sd, bn + '.running_var'),
                     _t(sd, bn + '.weight'), _t(sd, bn + '.bias'), False, 0.0, BN_EPS)
    if L['relu6']:
        x = F.relu6(x)                                                   # nn.ReLU6 (:41)
    return x


@torch.no_grad()
def mobilenet_v2_forward(sd: dict, x, return_features: bool = False):
    """x [B,3,120,120] float32 -> (param [B,62], pool [B,1280]) (+ per-feature outputs).

    sd: backbone state_dict without prefix (keys 'features.0.0.weight', ...).
    """
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.asarray(x, dtype=np.float32))
    feats = {}
    block_in = None
    cur_feature = -1
    for L in mbv2_layers():
        if L['feature'] != cur_feature:
            cur_feature = L['feature']
            block_in = x
        y = _conv_bn(sd, L, x)
        if L['residual']:
            y = block_in + y                                             # x + self.conv(x) (:70-72)
        x = y
        feats[L['key']] = x
    x = F.adaptive_avg_pool2d(x, 1).reshape(x.shape[0], -1)              # :179-180
    pool = x.clone()
    outs = [F.linear(x, _t(sd, n + '.weight'), _t(sd, n + '.bias')) for n, _ in HEADS]   # :184-186 (Dropout = id in eval)
    param = torch.cat(outs, dim=1)                                       # :188
    if return_features:
        return param, pool, feats
    return param, pool
