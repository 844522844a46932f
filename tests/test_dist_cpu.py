"""world_size-2 gloo tests (CPU) of the N>1 path: constant broadcast protocol + face sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


class _FakeModel:
    """Stands in for SynergyNet on CPU: same export/import protocol, no kernels."""

    def __init__(self, payload=None):
        self.device = torch.device('cpu')
        self.payload = payload

    def export_constants(self):
        return self.payload.clone()

    def import_constants(self, buf):
        self.payload = buf.clone()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from synergynet_amd.dist import broadcast_constants, shard_range
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        blob = torch.from_numpy(rng.integers(0, 256, 100003, dtype=np.uint8))
        m = _FakeModel(blob if rank == 0 else None)
        n = broadcast_constants(m, src=0)
        ok = (n == blob.numel()) and torch.equal(m.payload, blob)
        # face sharding: every face is processed exactly once, no exchange of results
        total = 8191
        lo, hi = shard_range(total, rank, world)
        mine = torch.zeros(total, dtype=torch.int64)
        mine[lo:hi] = 1
        dist.all_reduce(mine)            # test-only check of coverage
        ok = ok and bool((mine == 1).all())
        q.put((rank, ok, hi - lo))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert sorted(n for _, _, n in res) == [4095, 4096]


def test_shard_range_covers_everything():
    from synergynet_amd.dist import shard_range
    for total in (0, 1, 7, 1024, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
