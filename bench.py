#!/usr/bin/env python3
"""Throughput bench of the SynergyNet inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W           (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

One "step" = one pass of the hot path over one batch of synthetic crops per GPU:
uint8 crops [B,120,120,3] already resident in HBM -> MobileNetV2 -> 62 params ->
68 landmarks + 53215-vertex mesh (+ pose), all on device (BASELINE.json configs[2],
the configuration the metric "faces/sec (68-lmk + 53215-vert)" is quoted on; B = 1024
faces per GPU, i.e. configs[3]'s 8192 faces over 8 GPUs).  Weak scaling: every rank
processes its own shard of B faces; the only collective is the one-time RCCL broadcast
of the packed constants from rank 0 (outside the timed region).

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from synergynet_amd import synth                      # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
PEAK_HBM_GBS = 8000.0


class HipEvents:
    """hipEvent timing on an explicit stream (kernels are launched on torch's current stream)."""

    def __init__(self):
        from synergynet_amd import abi
        abi.lib()
        self.hip = ctypes.CDLL('libamdhip64.so.7')      # already loaded by torch: same runtime

    def create(self):
        e = ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, ctypes.c_void_p(stream)) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = ctypes.c_float()
        assert self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0
        return ms.value


def cpu_baseline(sd, pack, budget_s=20.0):
    """The reference's CPU path restated with the same torch-CPU/numpy ops (oracle/ = "port"),
    batched best case (BASELINE.md variant ii): forward(B) + batched dense+sparse reconstruction."""
    from oracle import backbone_torch, recon_numpy
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    b = recon_numpy.Basis(pack)
    Bc = 64
    x = synth.normalize_crops(synth.make_crops(Bc, seed=1))
    t0 = time.perf_counter()
    n = 0
    while True:
        p, _ = backbone_torch.mobilenet_v2_forward(sd, x)
        recon_numpy.reconstruct_vertex_62(b, p.numpy(), dense=False)
        recon_numpy.reconstruct_vertex_62(b, p.numpy(), dense=True)
        recon_numpy.predict_pose(b, p.numpy()[0], [0, 0, 120, 120, 1])
        n += Bc
        el = time.perf_counter() - t0
        if el > budget_s or n >= 64 * 40:
            break
    return dict(value=round(n / el, 2), unit='faces/s', cores=cores, kind='port',
                sample=f'{n} faces in batches of {Bc}: oracle torch-CPU MobileNetV2 forward + numpy 68-lmk and '
                       f'53215-vertex reconstruction, {cores} threads, {el:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='faces per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    assert torch.cuda.is_available(), 'bench.py needs MI355X GPUs'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from synergynet_amd.synergy3DMM import SynergyNet
    from synergynet_amd.dist import broadcast_constants
    pack = sd = None
    if rank == 0:
        pack, sd = synth.make_3dmm(), synth.make_backbone_state()
        model = SynergyNet(device=dev, pack=pack, backbone_state=sd)
    else:
        model = SynergyNet(device=dev, load_constants=False)
    if world > 1:
        broadcast_constants(model, src=0)          # one RCCL broadcast over xGMI, then no collectives

    B = args.batch
    crops = torch.from_numpy(synth.make_crops(B, seed=1000 + rank)).to(dev)      # this rank's shard
    rois = torch.from_numpy(synth.make_rois(B, seed=2000 + rank)).to(dev)
    lmk = torch.empty((B, 3, 68), dtype=torch.float32, device=dev)
    mesh = torch.empty((B, 3, 53215), dtype=torch.float32, device=dev)

    def step():
        param = model.forward_crops_u8(crops)
        model.reconstruct(param, roi=rois, dense=False, out=lmk)
        model.reconstruct(param, roi=rois, dense=True, out=mesh)
        model.predict_pose_batch(param, rois)

    for _ in range(args.warmup):
        step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # --- roofline of the dominant kernel class (pointwise fp32-MFMA convs), measured live with HIP events
    roof = None
    if rank == 0:
        ev = HipEvents()
        stream = torch.cuda.current_stream(dev).cuda_stream
        from synergynet_amd import abi
        lib = abi.lib()
        a, b_ = ev.create(), ev.create()
        param = torch.empty((B, 62), dtype=torch.float32, device=dev)
        reps = 5
        torch.cuda.synchronize()
        ev.record(a, stream)
        for _ in range(reps):
            abi.check(lib.syn_backbone_forward_u8(model._h, crops.data_ptr(), B, param.data_ptr(), None, ctypes.c_void_p(stream)))
        ev.record(b_, stream)
        bb_ms = ev.elapsed_ms(a, b_) / reps
        flops = lib.syn_backbone_flops_per_face() * B
        roof = dict(bound='mfma', kernel='backbone forward (53 launches; 34 pointwise fp32-MFMA GEMMs = 89.9% of flops)',
                    achieved=round(flops / (bb_ms * 1e-3) / 1e12, 3), peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s',
                    frac=round(flops / (bb_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), traffic=None,
                    ms_per_launch=round(bb_ms, 4), flops_per_launch=flops)

    if rank == 0:
        faces = B * world * args.steps
        out = dict(metric='faces/sec (120x120, 68-lmk + 53215-vert)', value=round(faces / el, 1), unit='faces/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(el / args.steps * 1e3, 4),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='MobileNetV2 120x120 uint8 crops -> 62 params -> 68 landmarks + 53215-vertex mesh '
                                        '+ pose, ROI affine, all on device (BASELINE configs[2]/[3])',
                               faces_per_gpu_per_step=B, global_batch=B * world, parallelism=f'face-shard x{world}',
                               collectives='one RCCL broadcast of packed constants at init, none in the timed region'),
                   roofline=roof)
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline(sd, pack)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
