"""GPU box: features.8 of the default schedule against the all-tiled one under simplified weights (which stage disagrees?)."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from synergynet_amd import abi, synth                       # noqa: E402
from synergynet_amd.synergy3DMM import SynergyNet            # noqa: E402

B = 768
pack = synth.make_3dmm()
x = torch.from_numpy(synth.normalize_crops(synth.make_crops(B, seed=5))).cuda()


def run(tag, edit):
    sd = {k: v.copy() for k, v in synth.make_backbone_state().items()}
    edit(sd)
    os.environ.pop('SYNERGY_HIP_EARLY_RM', None)
    m1 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    os.environ['SYNERGY_HIP_EARLY_RM'] = '0'
    m0 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    os.environ.pop('SYNERGY_HIP_EARLY_RM', None)
    a = torch.empty((B, 8, 8, 64), dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    abi.check(abi.lib().syn_debug_feature(m1._h, x.data_ptr(), B, 8, a.data_ptr(), None))
    abi.check(abi.lib().syn_debug_feature(m0._h, x.data_ptr(), B, 8, b.data_ptr(), None))
    torch.cuda.synchronize()
    d = (a - b).abs().cpu().numpy()
    bad = d > 1e-4 * b.abs().max().item()
    print(f'{tag:28s} max abs diff {d.max():.3e} ref max {b.abs().max().item():.3e} bad {bad.mean():.4f}  by y {np.round(bad.mean((0, 2, 3)), 2)} by x {np.round(bad.mean((0, 1, 3)), 2)}')


def dw_only(taps):
    def edit(sd):
        w = sd['features.8.conv.1.0.weight']
        keep = np.zeros_like(w)
        for (ky, kx) in taps:
            keep[:, 0, ky, kx] = w[:, 0, ky, kx]
        sd['features.8.conv.1.0.weight'] = keep
    return edit


def hid_only(lo, hi):
    def edit(sd):
        dw_only([(1, 1)])(sd)
        w = sd['features.8.conv.2.weight']
        keep = np.zeros_like(w)
        keep[:, lo:hi] = w[:, lo:hi]
        sd['features.8.conv.2.weight'] = keep
    return edit


run('normal', lambda sd: None)
run('dw centre', dw_only([(1, 1)]))
run('dw centre+left', dw_only([(1, 1), (1, 0)]))
run('dw centre+right', dw_only([(1, 1), (1, 2)]))
run('dw centre+up', dw_only([(1, 1), (0, 1)]))
run('dw centre+down', dw_only([(1, 1), (2, 1)]))
run('centre, hidden 0..32', hid_only(0, 32))
run('centre, hidden 0..1', hid_only(0, 1))
run('centre, hidden 4..5', hid_only(4, 5))
run('centre, hidden 16..17', hid_only(16, 17))
run('centre, hidden 32..64', hid_only(32, 64))
