import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
pack, sd = synth.make_3dmm(1), synth.make_backbone_state(2)
crops = torch.from_numpy(synth.make_crops(64, seed=5)).cuda()
for rm in ('2047', '1023'):
    os.environ['SYNERGY_HIP_EARLY_RM'] = rm
    m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    base = m.forward_crops_u8(crops[:2].contiguous())
    for B in (3, 4, 5, 8, 31, 32, 33, 64):
        g = m.forward_crops_u8(crops[:B].contiguous())[:2]
        again = m.forward_crops_u8(crops[:B].contiguous())[:2]
        print('EARLY_RM', rm, 'B', B, 'equal to B=2:', bool(torch.equal(g, base)), 'max diff', float((g - base).abs().max()), 'rerun equal:', bool(torch.equal(g, again)))
