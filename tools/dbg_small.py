"""GPU box: default schedule vs the all-tiled schedule (SYNERGY_HIP_EARLY_RM=0) at small / medium batches: parameters of distinct faces."""
import os, sys
import numpy as np
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
pack, sd = synth.make_3dmm(), synth.make_backbone_state()
m1 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
os.environ['SYNERGY_HIP_EARLY_RM'] = '0'
m0 = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
for B in [int(x) for x in sys.argv[1:]] or [32, 33, 100, 128, 255, 511, 767]:
    c = torch.from_numpy(synth.make_crops(B, seed=100 + B)).cuda()
    a, b = m1.forward_crops_u8(c).cpu().numpy().astype(np.float64), m0.forward_crops_u8(c).cpu().numpy().astype(np.float64)
    pf = np.abs(a - b).max(1) / np.abs(b).max(1)
    a2 = m1.forward_crops_u8(torch.roll(c, 5, 0)).cpu().numpy()
    print(f'B={B}: max per-face rel diff {pf.max():.2e} (face {pf.argmax()}); position-independent: {np.array_equal(np.roll(a2, -5, 0), a.astype(np.float32))}')
