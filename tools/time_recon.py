"""GPU box: reconstruction alone (dense; pitched output, or the reference's packed [B,3,53215] with a third argument `packed`), ms per call at
batch B.  usage: python tools/time_recon.py [B] [iters] [packed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
it = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(), backbone_state=synth.make_backbone_state())
p = torch.from_numpy(synth.make_params(B, seed=5)).cuda(); roi = torch.from_numpy(synth.make_rois(B, seed=6)).cuda()
out = m.empty_vertices(B)
if len(sys.argv) > 3 and sys.argv[3] == 'packed':
    out = torch.empty((B, 3, 53215), dtype=torch.float32, device='cuda:0')
for _ in range(20): m.reconstruct(p, roi, dense=True, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(3):
    e0.record()
    for _ in range(it): m.reconstruct(p, roi, dense=True, out=out)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / it)
print(('packed ' if len(sys.argv) > 3 else 'pitched ') + 'recon B=%d  %.4f ms  %.2f TB/s of mesh writes  knobs=%s' % (B, best, B * 3 * 53215 * 4 / best / 1e9, os.environ.get('SYNERGY_HIP_TEST_KNOBS', '-')))
