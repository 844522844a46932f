"""GPU box: the headline step with host buffers on both sides (pinned): H2D of the uint8 crops + ROIs, forward, landmarks + mesh + pose,
D2H of landmarks and mesh.  The C ABI takes device pointers -- this is the rate a caller with host data would see, never bench.py's `value`."""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(), backbone_state=synth.make_backbone_state())
h_crops = torch.from_numpy(synth.make_crops(B, seed=3)).pin_memory()
h_roi = torch.from_numpy(synth.make_rois(B, seed=4)).pin_memory()
d_crops = torch.empty_like(h_crops, device='cuda'); d_roi = torch.empty_like(h_roi, device='cuda')
mesh = m.empty_vertices(B)
h_mesh = torch.empty(mesh.shape, dtype=torch.float32).pin_memory()
h_lmk = torch.empty((B, 3, 68), dtype=torch.float32).pin_memory()
def step(with_mesh=True):
    d_crops.copy_(h_crops, non_blocking=True); d_roi.copy_(h_roi, non_blocking=True)
    p = m.forward_crops_u8(d_crops)
    lmk = m.reconstruct(p, d_roi, dense=False)
    h_lmk.copy_(lmk, non_blocking=True)
    if with_mesh:
        m.reconstruct(p, d_roi, dense=True, out=mesh)
        h_mesh.copy_(mesh, non_blocking=True)
for wm in (True, False):
    for _ in range(3): step(wm)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 10
    for _ in range(n): step(wm)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f'B={B} host-to-host, {"landmarks + mesh" if wm else "landmarks only"}: {dt*1e3:.2f} ms/step = {B/dt/1e3:.1f} k faces/s'
          + (f' (mesh download {mesh.numel()*4/1e6:.0f} MB at {mesh.numel()*4/dt/1e9:.1f} GB/s incl. everything)' if wm else ''))
