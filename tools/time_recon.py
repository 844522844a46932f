"""GPU box: time syn_reconstruct (dense, B faces) under both schedules."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NV = int(sys.argv[2]) if len(sys.argv) > 2 else 53215
pack = synth.make_3dmm(n_vert=NV); sd = synth.make_backbone_state()
for f in (1, 2):
    os.environ['SYNERGY_HIP_FUSION'] = str(f)
    m = SynergyNet(device='cuda:0', pack=pack, backbone_state=sd)
    p = torch.from_numpy(synth.make_params(B, seed=5)).cuda(); roi = torch.from_numpy(synth.make_rois(B, seed=6)).cuda()
    out = m.empty_vertices(B) if os.environ.get('PACKED') != '1' else torch.empty((B, 3, NV), dtype=torch.float32, device='cuda')
    lmk = torch.empty((B, 3, 68), dtype=torch.float32, device='cuda')
    for dense, buf, name in ((False, lmk, 'landmarks (host-call floor)'), (True, out, 'dense')):
        for _ in range(3): m.reconstruct(p, roi, dense=dense, out=buf)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): m.reconstruct(p, roi, dense=dense, out=buf)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f'fusion={f} B={B} nv={NV} {name} recon {ms*1e3:.1f} us  -> {B*3*(NV if dense else 68)*4/ms/1e9:.2f} TB/s of output')
