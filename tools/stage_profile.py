"""Per-stage s_memtime profile of the fused block kernels (GPU box): python tools/stage_profile.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synergynet_amd import abi, synth
from synergynet_amd.synergy3DMM import SynergyNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = SynergyNet(device='cuda:0', pack=synth.make_3dmm(n_vert=640), backbone_state=synth.make_backbone_state())
x = torch.from_numpy(synth.normalize_crops(synth.make_crops(B, seed=1))).cuda()
names = ['stage0', 'expand', 'bar1', 'dw', 'bar2', 'project', 'epilog']
print(f'{"feature":>8s} ' + ' '.join(f'{n:>9s}' for n in names) + '     total  (shader cycles per tile, view of wave 0)')
for f in ([int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else (2, 3, 4, 5, 7, 8, 11, 12, 14, 15, 17)):  # features.8-13 (fused_block_lb.hip): columns = staging, expand, depthwise, project, exchange+store, averaged over all waves
    out = (C.c_ulonglong * 32)()
    abi.check(abi.lib().syn_debug_profile_block(m._h, x.data_ptr(), B, f, out))
    n = max(out[7], 1)
    v = [out[i] / n for i in range(7)]
    print(f'{f:8d} ' + ' '.join(f'{t:9.1f}' for t in v) + f' {sum(v):9.1f}   wgs={out[7]}')
    if any(out[8:]):
        print('         busy cycles per step by wave id of the workgroup (row-marching kernels): ' + ' '.join(f'{out[8 + i] / n:.0f}' for i in range(24) if out[8 + i]))
