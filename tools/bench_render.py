"""GPU box: throughput of the mesh consumers (normals + Phong vertex colours + z-buffer rasterisation + alpha blend) on
full-size meshes (53215 vertices, 105840 triangles) drawn into a 450x450 frame, device-resident, next to the CPU oracle
(C restatement of the reference's C++ rasteriser + numpy lighting = what the reference runs, one host thread).
usage: python tools/bench_render.py [F ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth, sim3dr
from synergynet_amd.synergy3DMM import SynergyNet
from oracle import sim3dr as osim          # CPU-baseline leg only (same role as bench.py's cpu_baseline): timed beside, never inside, the GPU path

Fs = [int(a) for a in sys.argv[1:]] or [1, 8, 64]
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(n_vert=640), backbone_state=synth.make_backbone_state())
tri = synth.make_grid_topology(n_vert=53215, n_tri=105840)
m.triangles = torch.from_numpy(np.ascontiguousarray(tri.T).astype(np.int64))
H = W = 450
img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
img_t = torch.from_numpy(img).cuda()
for F in Fs:
    meshes = synth.make_face_meshes(F, n_vert=53215, height=H, width=W, seed=5)
    mt = torch.from_numpy(meshes).cuda()
    for _ in range(2): sim3dr.render_batch(m, img_t, mt)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    a.record()
    for _ in range(n): sim3dr.render_batch(m, img_t, mt)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    t0 = time.perf_counter()
    nf = min(F, 8)
    osim.render_overlay(img, [meshes[f] for f in range(nf)], tri)
    cpu = (time.perf_counter() - t0) / nf
    # compulsory bytes per mesh: vertices 639 KB read (x3 stages) + normals/light 2 x 639 KB written+read; the frame is 608 KB
    print(f'F={F:3d}: GPU {ms*1e3:8.1f} us per frame = {F/ms*1e3:9.0f} meshes/s   |  CPU oracle {cpu*1e3:6.1f} ms per mesh = {1/cpu:5.1f} meshes/s (1 thread)')
