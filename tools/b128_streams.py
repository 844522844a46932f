"""B = 128 landmarks-only batches (BASELINE configs[1]) submitted round-robin to N independent replicas (synergynet_amd.streams.ReplicaRing:
one handle + one HIP stream each): what a serving process that always has several small batches in flight gets out of one GPU, next to
the one-batch-at-a-time step.  `--json`: one JSON object on stdout (bench.py runs this in a process of its own so that the extra streams
never exist in the process that times the headline)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from synergynet_amd import synth  # noqa: E402
from synergynet_amd.streams import ReplicaRing  # noqa: E402
from synergynet_amd.synergy3DMM import SynergyNet  # noqa: E402


def main():
    as_json = '--json' in sys.argv
    counts = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 3, 4]
    pack, sd = synth.make_3dmm(), synth.make_backbone_state()
    dev = torch.device('cuda:0')
    B = 128
    crops = torch.from_numpy(synth.make_crops(B, seed=3)).to(dev)
    rois = torch.from_numpy(synth.make_rois(B, seed=5)).to(dev)
    out = {}
    ref = None
    for n in counts:
        ring = ReplicaRing(lambda: SynergyNet(device='cuda:0', pack=pack, backbone_state=sd), n)
        for _ in range(40):
            res = ring.submit(crops, rois)
        ring.wait()
        if ref is None:
            ref = res[1].clone()
        same = bool(torch.equal(res[1], ref))
        steps = 600
        t = time.perf_counter()
        for _ in range(steps):
            ring.submit(crops, rois)
        ring.wait()
        dt = time.perf_counter() - t
        out[str(n)] = dict(faces_s=round(B * steps / dt, 1), ms_per_batch=round(dt / steps * 1e3, 4), same_bits_as_one_replica=same)
        if not as_json:
            print(f'{n} replica(s): {B * steps / dt:10.0f} faces/s  {dt / steps * 1e3:.4f} ms per batch of {B}  same bits: {same}')
        del ring
    if as_json:
        print(json.dumps(out))


if __name__ == '__main__':
    main()
