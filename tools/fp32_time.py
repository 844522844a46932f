import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(1), backbone_state=synth.make_backbone_state(2))
B = 1024
crops = synth.make_crops(B, seed=3)
xf = torch.from_numpy(synth.normalize_crops(crops)).cuda(); cu = torch.from_numpy(crops).cuda()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
t(lambda: m.forward_crops_u8(cu), 200)            # clocks up first: the first second of work runs ~8 % slower on every kernel
for _ in range(3):                                  # ... and alternate
    print('fp32 ingest backbone ms', round(t(lambda: m.forward_test(xf)), 4), ' u8 ingest', round(t(lambda: m.forward_crops_u8(cu)), 4))
