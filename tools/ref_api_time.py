"""Where the reference-API-shaped step (bench.py extra.reference_api_step) spends its time: per call the host enqueue time (perf_counter
around the loop, queue drained before) and the GPU time per step (HIP events), B = 1024.  usage: python tools/ref_api_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(1), backbone_state=synth.make_backbone_state(2))
B = 1024
crops = synth.make_crops(B, seed=3)
xf = torch.from_numpy(synth.normalize_crops(crops)).cuda()
rois = torch.from_numpy(synth.make_rois(B, seed=4)).cuda() if hasattr(synth, 'make_rois') else None
lmk = torch.empty((B, 3, 68), dtype=torch.float32, device='cuda')
packed = torch.empty((B, 3, m._n_vert), dtype=torch.float32, device='cuda')
p0 = m.forward_test(xf)


def both(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, e0.elapsed_time(e1) / n, (t2 - t0) / n * 1e3


def step():
    p = m.forward_test(xf)
    m.reconstruct(p, roi=rois, dense=False, out=lmk)
    m.reconstruct(p, roi=rois, dense=True, out=packed)
    m.predict_pose_batch(p, rois)


both(lambda: m.forward_test(xf), 300)               # clocks up
for name, fn in [('forward_test', lambda: m.forward_test(xf)),
                 ('reconstruct lmk', lambda: m.reconstruct(p0, roi=rois, dense=False, out=lmk)),
                 ('reconstruct packed', lambda: m.reconstruct(p0, roi=rois, dense=True, out=packed)),
                 ('predict_pose_batch', lambda: m.predict_pose_batch(p0, rois)),
                 ('whole step', step), ('whole step', step)]:
    h, g, w = both(fn)
    print('%-20s host enqueue %.4f ms   gpu (events) %.4f ms   wall incl. drain %.4f ms' % (name, h, g, w))
