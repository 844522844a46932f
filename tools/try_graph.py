"""GPU box: does a HIP graph (torch.cuda.CUDAGraph capture of the library's launches) cut the launch-bound small-batch step?"""
import os, sys, time
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from synergynet_amd import synth
from synergynet_amd.synergy3DMM import SynergyNet
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(), backbone_state=synth.make_backbone_state())
for B in (1, 8, 128):
    crops = torch.from_numpy(synth.make_crops(B, seed=3)).cuda()
    roi = torch.from_numpy(synth.make_rois(B, seed=4)).cuda()
    def step():
        p = m.forward_crops_u8(crops)
        return p, m.reconstruct(p, roi, dense=False)
    for _ in range(5): out = step()
    torch.cuda.synchronize()
    def timeit(fn, n=300):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n)
        return best
    t_eager = timeit(step)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): step()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            gout = step()
        torch.cuda.synchronize()
        t_graph = timeit(g.replay)
        ok = torch.equal(gout[0], out[0]) and torch.equal(gout[1], out[1])
        print(f'B={B}: eager {t_eager*1e3:.1f} us/step, graph replay {t_graph*1e3:.1f} us/step, same results: {ok}')
    except Exception as ex:
        print(f'B={B}: eager {t_eager*1e3:.1f} us/step, graph capture failed: {type(ex).__name__}: {str(ex)[:300]}')
