"""Multi-GPU plumbing for the face-sharded hot path.

The path is embarrassingly parallel over faces: one process per GPU, each holding a full
replica of the packed constants (folded backbone ~9 MB + MFMA-ordered 3DMM basis ~33 MB) and
processing a contiguous shard of the batch.  The only communication is ONE broadcast of the
packed constants from the rank that loaded them (RCCL over xGMI when the backend is "nccl");
results stay on the GPU that produced them - no gather, no reduction.

The reference's only multi-GPU construct is nn.DataParallel in the training / benchmark
scripts (main_train.py:176, benchmark.py:112), which re-broadcasts the module on every call.
"""
from __future__ import annotations

import torch


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split of n_items over world ranks; the first (n_items % world) ranks get one extra."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_constants(model, src: int = 0, group=None):
    """Rank `src` exports its packed constants; every other rank imports them.

    `model` needs .device, .export_constants() -> uint8 tensor on that device and
    .import_constants(tensor) (synergynet_amd.SynergyNet provides them)."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    n = torch.zeros(1, dtype=torch.int64, device=model.device)
    buf = None
    if rank == src:
        buf = model.export_constants()
        n[0] = buf.numel()
    dist.broadcast(n, src, group=group)
    if rank != src:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=model.device)
    dist.broadcast(buf, src, group=group)
    if rank != src:
        model.import_constants(buf)
    return int(n.item())
