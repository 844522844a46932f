"""GPU box: ResNet-50 at B = 9 with a set of convolutions forced onto the exact kernel (SYNERGY_HIP_TEST_KNOBS=resnet_exact_mask=0x...) vs the oracle."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
from oracle import resnet_torch
sd = synth.make_resnet50_state()
crops = synth.make_crops(9, seed=4)
want = resnet_torch.resnet50_forward(sd, synth.normalize_crops(crops))[0].numpy()[:, :62]
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(n_vert=640), backbone_state=sd, arch='resnet50')
got = m.forward_crops_u8(torch.from_numpy(crops).cuda()).cpu().numpy()
print('%.3e' % (np.abs(got - want).max(axis=1) / np.abs(want).max(axis=1)).max())
'''
for mask in sys.argv[1:]:
    env = dict(os.environ, SYNERGY_HIP_TEST_KNOBS='resnet_exact_mask=0x' + mask)
    r = subprocess.run([sys.executable, '-c', SCRIPT, ROOT], env=env, capture_output=True, text=True)
    print(mask, r.stdout.strip(), r.stderr.strip()[-300:] if r.returncode else '')
