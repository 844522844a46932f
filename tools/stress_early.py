"""GPU box: compare features.2-4 of the default (fp16x2) schedule against the fp32-MFMA schedule on many random batches."""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import abi, synth
from synergynet_amd.synergy3DMM import SynergyNet
pack = synth.make_3dmm(n_vert=640); sd = synth.make_backbone_state()
def mk(f):
    os.environ['SYNERGY_HIP_FUSION'] = str(f)
    return SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
m1, m2 = mk(1), mk(2)
shapes = {2: (30, 24), 3: (30, 24), 4: (15, 32)}
worst = 0.0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    B = [1, 3, 17, 64, 257, 300][it % 6]
    x = torch.from_numpy(synth.normalize_crops(synth.make_crops(B, seed=100 + it))).cuda()
    for f, (hw, c) in shapes.items():
        a = torch.empty((B, hw, hw, c), device='cuda'); b = torch.empty_like(a)
        abi.check(abi.lib().syn_debug_feature(m1._h, x.data_ptr(), B, f, a.data_ptr(), None))
        abi.check(abi.lib().syn_debug_feature(m2._h, x.data_ptr(), B, f, b.data_ptr(), None))
        torch.cuda.synchronize()
        e = float((a - b).abs().max() / a.abs().max())
        worst = max(worst, e)
        if not e < 1e-5:
            print(f'MISMATCH it={it} B={B} feature={f} rel={e:.3e} bad_px={int(((a-b).abs().amax(dim=3) > 1e-4 * a.abs().max()).sum())}')
print('worst rel', worst)
