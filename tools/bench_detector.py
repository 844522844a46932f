"""GPU box: FaceBoxes detector latency per frame (syn_detect: preprocessing, 31 convolutions, pools, decode, sort, NMS) next to
the torch-CPU oracle.   usage: python tools/bench_detector.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import synth
from synergynet_amd.faceboxes import FaceBoxes
from oracle import faceboxes_torch as ofb  # CPU-baseline leg only (same role as bench.py's cpu_baseline): timed beside, never inside, the GPU path
sd = synth.make_faceboxes_state()
det = FaceBoxes(state_dict=sd)
for hw in ((300, 420), (720, 1080), (1080, 1920)):
    frame = synth.make_frame(*hw, seed=1)
    ft = torch.from_numpy(frame).cuda()
    for _ in range(3): det.detect_all(ft)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): d = det.detect_all(ft)
    torch.cuda.synchronize(); gpu = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); w = ofb.detect(sd, frame, return_all=True); cpu = time.perf_counter() - t0
    print(f'{hw[0]}x{hw[1]}: GPU {gpu*1e3:6.2f} ms/frame ({1/gpu:6.0f} fps, {d.shape[0]} dets)   torch-CPU oracle {cpu*1e3:7.1f} ms ({torch.get_num_threads()} threads)')
