"""GPU box: ResNet-50 (BASELINE configs[4]) backbone time at B faces, per-kernel via torch events around forward_crops_u8.
usage: python tools/time_resnet.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(n_vert=640), backbone_state=synth.make_resnet50_state(), arch='resnet50')
c = torch.from_numpy(synth.make_crops(B, seed=1)).cuda()
for _ in range(3): m.forward_crops_u8(c)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): m.forward_crops_u8(c)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
from synergynet_amd import abi
fl = abi.lib().syn_resnet50_flops_per_face() * B
print(f'B={B} SYNERGY_HIP_EARLY_RM={os.environ.get("SYNERGY_HIP_EARLY_RM", "default")}: backbone {ms:.3f} ms  {B / ms * 1e3:.0f} faces/s  {fl / ms / 1e9:.1f} TF')
