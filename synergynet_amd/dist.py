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


def broadcast_constants_rccl(model, nccl_comm: int, root: int = 0):
    """The same hand-off through the library's own collective (include/synergy_hip.h syn_bcast_constants): `nccl_comm` is the
    address of an ncclComm_t of the RCCL instance this process holds (a C host's, or one made through ctypes on torch's
    librccl.so.1); every rank calls it, rank `root` is the one that loaded.  For hosts without torch.distributed."""
    import ctypes as C

    from . import abi
    with torch.cuda.device(model.device):
        s = torch.cuda.current_stream(model.device).cuda_stream
        abi.check(abi.lib().syn_bcast_constants(model._h, C.c_void_p(nccl_comm), int(root), C.c_void_p(s)))
    model._after_import()


def pack_constants_host(pack=None, backbone_state=None, arch: str = 'mobilenet_v2', data_dir=None):
    """The constants blob of `SynergyNet(...).export_constants()` built WITHOUT a device (syn_pack_constants_host): a uint8
    numpy array [header 256 B | folded backbone | MFMA-ordered basis], byte-identical to the device export after the same
    loads (tests/test_gpu_parity.py).  `pack` / `data_dir`: the 3DMM assets as for SynergyNet (None, None -> no basis);
    `backbone_state`: plain state_dict of the backbone (None -> no backbone)."""
    import ctypes as C

    import numpy as np

    from . import abi
    from .params import ParamsPack
    from .synergy3DMM import flatten_backbone
    lib = abi.lib()
    a = {'mobilenet_v2': 0, 'resnet50': 1}[arch]
    flat = None
    if backbone_state is not None:
        sd = {k: v for k, v in backbone_state.items() if not k.endswith('num_batches_tracked')}
        flat = flatten_backbone(sd, '', arch)
    n_vert = n_lmk = 0
    arrs = [None] * 6
    if pack is not None or data_dir is not None:
        pp = ParamsPack(data_dir=data_dir, pack=pack)
        f32 = lambda x: np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        kp = np.ascontiguousarray(np.asarray(pp.keypoints), dtype=np.int64)
        arrs = [f32(pp.w_shp), f32(pp.w_exp), f32(pp.u).reshape(-1), f32(pp.param_mean).reshape(-1), f32(pp.param_std).reshape(-1), kp]
        if arrs[3].size < 62 or arrs[4].size < 62:
            raise RuntimeError('param_mean/param_std shorter than 62')
        n_vert, n_lmk = arrs[0].shape[0] // 3, kp.size // 3
    n = int(lib.syn_pack_constants_host_bytes(a, int(flat is not None), n_vert, n_lmk))
    out = np.empty(n, dtype=np.uint8)
    ptr = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
    abi.check(lib.syn_pack_constants_host(a, ptr(flat), 0 if flat is None else flat.size, *[ptr(x) for x in arrs], n_lmk, n_vert,
                                          out.ctypes.data_as(C.c_void_p), n))
    return out


def check_constants_host(blob) -> dict:
    """Vet a host copy of a constants blob exactly like syn_import_constants would (C side, syn_check_constants_host) and
    return its parsed header (Python side, synergy3DMM.parse_constants_header); raises on a blob the import would refuse."""
    import ctypes as C

    import numpy as np

    from . import abi
    from .synergy3DMM import parse_constants_header
    b = np.ascontiguousarray(np.asarray(blob, dtype=np.uint8))
    abi.check(abi.lib().syn_check_constants_host(b.ctypes.data_as(C.c_void_p), b.size))
    return parse_constants_header(b[:256].tobytes())
