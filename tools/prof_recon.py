"""GPU box: per-stage ticks of recon_f16_kernel (SYNERGY_HIP_TEST_KNOBS=recon_prof=1 makes launch_reconstruct_f16 run the instrumented variant
and print averages per workgroup to stderr; s_memtime ticks are 10 ns)."""
import os, sys
os.environ['SYNERGY_HIP_TEST_KNOBS'] = 'recon_prof=1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(), backbone_state=synth.make_backbone_state())
p = torch.from_numpy(synth.make_params(B, seed=5)).cuda(); roi = torch.from_numpy(synth.make_rois(B, seed=6)).cuda()
out = m.empty_vertices(B)
for _ in range(3):
    m.reconstruct(p, roi, dense=True, out=out)
    sys.stderr.write('--\n')
