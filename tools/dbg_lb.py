"""GPU box: features.7-17 of the default schedule against the all-tiled schedule (SYNERGY_HIP_EARLY_RM=0) at B faces."""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from synergynet_amd import abi, synth                       # noqa: E402
from synergynet_amd.synergy3DMM import SynergyNet            # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 768
pack, sd = synth.make_3dmm(), synth.make_backbone_state()
m1 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
os.environ['SYNERGY_HIP_EARLY_RM'] = '0'
m0 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
x = torch.from_numpy(synth.normalize_crops(synth.make_crops(B, seed=5))).cuda()
shapes = {7: (8, 64), 8: (8, 64), 9: (8, 64), 10: (8, 64), 11: (8, 96), 12: (8, 96), 13: (8, 96), 14: (4, 160), 15: (4, 160), 16: (4, 160), 17: (4, 320)}
for f, (h, c) in shapes.items():
    a = torch.empty((B, h, h, c), dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    abi.check(abi.lib().syn_debug_feature(m1._h, x.data_ptr(), B, f, a.data_ptr(), None))
    abi.check(abi.lib().syn_debug_feature(m0._h, x.data_ptr(), B, f, b.data_ptr(), None))
    torch.cuda.synchronize()
    d = (a - b).abs().cpu().numpy()
    ref = b.abs().max().item()
    print(f'features.{f}: max abs diff {d.max():.3e} (ref max {ref:.3e})')
    if d.max() > 1e-4 * ref:
        bad = d > 1e-4 * ref
        print('   bad fraction', bad.mean(), 'by face[:4]', bad.reshape(B, -1).mean(1)[:4], '\n   by y', bad.mean((0, 2, 3)), '\n   by x', bad.mean((0, 1, 3)),
              '\n   by channel', np.round(bad.mean((0, 1, 2)), 2))
        break
