"""Two-stream schedule of the hot path for a stream of batches.

The backbone of a batch is bound by vector/matrix issue, its reconstruction by HBM writes (654 MB of mesh per 1024 faces):
run back to back they leave the other resource idle.  `OverlappedPipeline` puts the reconstruction (+ landmarks + pose) of
batch i on a second HIP stream so that it runs beside the backbone of batch i+1; events order the two streams and keep the
parameters of a batch alive until its reconstruction has read them (at most two batches in flight).  Results are identical
to the sequential calls -- the same kernels on the same inputs; only their placement in time changes (+2.5 % throughput at
B = 1024).  The two stages use separate scratch allocations of the handle (csrc/synergy_abi.hip: `ws` activations, `rec`
reconstruction records), each of which only grows -- behind a device-wide synchronisation -- so batches of ANY sizes may follow
each other (tests/test_gpu_parity.py::test_two_stream_pipeline_with_varying_batch_sizes).  All reconstruction calls of one
handle must stay on ONE stream at a time (they share `rec`); that is what this class does.
"""
from __future__ import annotations

import torch


_REC_STREAMS = {}       # (device index, priority) -> the one reconstruction stream of this process


def reconstruction_stream(dev, priority=-1):
    """The second HIP stream, ONE per (device, priority) for the whole process: every extra stream is another hardware queue
    for the command processor to rotate through, and a few pipelines each with a stream of its own were measured 30 % slower
    (ResNet-50 B = 512: 11.2 -> 14.7 ms/step once a second high-priority stream existed) than the same work on a shared one."""
    key = (torch.device(dev).index, priority)
    if key not in _REC_STREAMS:
        _REC_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return _REC_STREAMS[key]


class OverlappedPipeline:
    def __init__(self, model, overlap=True, rec_priority=-1):
        self.model = model
        self.dev = model.device
        self.s_main = torch.cuda.current_stream(self.dev)
        # high priority: the short store-bound kernels get their workgroups in as soon as a backbone workgroup retires (+0.3 %)
        self.s_rec = reconstruction_stream(self.dev, rec_priority) if overlap else self.s_main
        self._inflight = [None, None]              # per parity: (tensors kept alive, event "second stage done")
        self._n = 0
        self._tail_stream = None
        self.last_done = None

    def submit(self, crops_u8, rois, lmk_out=None, mesh_out=None, dense=True):
        """Enqueue one batch: uint8 crops [B,120,120,3] and rois [B,5] (device tensors).  Returns (param, lmk, mesh, (angles,
        t3d)) device tensors that are valid after `wait()` (or after synchronising with the event in `.last_done`).
        dense=False: landmarks + pose only (BASELINE configs[1]); mesh is None."""
        m, k = self.model, self._n & 1
        self._n += 1
        if self._inflight[k] is not None:
            self.s_main.wait_event(self._inflight[k][1])
        with torch.cuda.stream(self.s_main):
            param = m.forward_crops_u8(crops_u8)
            ready = torch.cuda.Event()
            ready.record(self.s_main)
        # landmarks + pose only (three kernels of ~5 us): nothing to overlap, and the hop to the second stream costs as much as they take
        # (configs[1], B = 128: 0.335 -> 0.331 ms) -- they stay behind the backbone on its stream
        s_tail = self.s_rec if dense else self.s_main
        with torch.cuda.stream(s_tail):
            s_tail.wait_event(ready)
            if self._tail_stream is not s_tail and self.last_done is not None:
                s_tail.wait_event(self.last_done)      # (a dense batch's tail may still use the handle's reconstruction records on the other stream)
            self._tail_stream = s_tail
            lmk, pose = m.landmarks_and_pose(param, roi=rois, out=lmk_out)        # one launch (round 5; was prologue + contraction + pose)
            mesh = m.reconstruct(param, roi=rois, dense=True, out=mesh_out) if dense else None
            done = torch.cuda.Event()
            done.record(s_tail)
        self._inflight[k] = ((param, lmk, mesh, pose, crops_u8, rois), done)
        self.last_done = done
        return param, lmk, mesh, pose

    def wait(self):
        for slot in self._inflight:
            if slot is not None:
                slot[1].synchronize()


class ReplicaRing:
    """N independent replicas -- a handle (its own workspace and copy of the constants, ~104 MB) and a HIP stream each -- that take
    batches round-robin.  One small batch (BASELINE configs[1]: B = 128, landmarks + pose) is a chain of ~30 dependent launches of 5-35 us
    that each use part of the chip: 0.39 ms whether it holds 1 face or 128.  A process that always has several such batches in flight
    (a server draining a request queue) gets their launches interleaved by the hardware: measured 331 k faces/s with one replica,
    506 k with two, 558 k with three, 575 k with four (tools/b128_streams.py, 0.223 ms per batch of 128).  Latency per batch is unchanged;
    results are the same bits as a lone replica's (same kernels, same inputs).
    Large batches gain too: at B = 1024 with the mesh, TWO replicas alternating beat one handle with a reconstruction stream
    (OverlappedPipeline) by 3.6-4 % on the same box (1.070 -> 1.032 ms per batch) -- the whole tail of batch i (the store-bound
    reconstruction, the short last rounds of its kernels) runs beside batch i + 1 -- as long as the process owns no other streams: every
    further HIP stream that has ever been used costs all of them (1.052 -> 1.121 ms with six more in the process).  bench.py's headline
    uses exactly this (--overlap 2)."""

    def __init__(self, make_model=None, n=4, models=None):
        self.models = list(models) if models is not None else [make_model() for _ in range(n)]
        dev = self.models[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.models]
        self._k = 0

    def submit(self, crops_u8, rois, dense=False, lmk_out=None, mesh_out=None):
        """Enqueue one batch on the next replica; returns (param, lmk, mesh | None, (angles, t3d), done_event).  lmk_out / mesh_out: result
        buffers of THIS batch (a buffer must not be handed to a later batch before this batch's event has fired)."""
        i = self._k % len(self.models)
        self._k += 1
        m, st = self.models[i], self.streams[i]
        st.wait_stream(torch.cuda.current_stream(m.device))          # the inputs were produced on the caller's stream
        with torch.cuda.stream(st):
            param = m.forward_crops_u8(crops_u8)
            lmk, pose = m.landmarks_and_pose(param, roi=rois, out=lmk_out)
            mesh = m.reconstruct(param, roi=rois, dense=True, out=mesh_out) if dense else None
            done = torch.cuda.Event()
            done.record(st)
        for t in (crops_u8, rois):              # (numpy inputs / roi=None are staged by the calls above into tensors of their own)
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(st)
        return param, lmk, mesh, pose, done

    def wait(self):
        for st in self.streams:
            st.synchronize()
