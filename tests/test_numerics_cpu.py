"""CPU tests of the load-time range analysis of the fp16x2 schedule (csrc/synergy_abi.hip analyze_mbv2_ranges): the verdict is
computed by the device-free packer (syn_pack_constants_host) and rides in the constants blob, so it can be checked here without a
GPU.  The GPU side (tests/test_gpu_numerics.py) holds the kernels to the oracle on the same adversarial checkpoints."""
import ctypes as C

import numpy as np
import pytest

import adversarial as adv
from synergynet_amd import synth
from synergynet_amd.build import LIB
from synergynet_amd.synergy3DMM import flatten_backbone


@pytest.fixture(scope='module')
def lib():
    l = C.CDLL(LIB)
    l.syn_pack_constants_host_bytes.restype = C.c_size_t
    l.syn_pack_constants_host_bytes.argtypes = [C.c_int] * 4
    l.syn_pack_constants_host.argtypes = [C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    return l


def verdict(lib, sd):
    """(unsafe1 mask, unsafe16 mask, input bound per feature, min weighted-mean bound, weight criterion) from the blob's tail."""
    flat = flatten_backbone(sd)
    n = lib.syn_pack_constants_host_bytes(0, 1, 0, 0)
    buf = np.zeros(n, dtype=np.uint8)
    assert lib.syn_pack_constants_host(0, flat.ctypes.data, flat.size, None, None, None, None, None, None, 0, 0, buf.ctypes.data, n) == 0
    tail = buf[-256:]
    u = tail.view(np.uint32)
    f = tail.view(np.float32)
    return int(u[0]), int(u[1]), f[2:22].copy(), f[22:42].copy(), f[42:62].copy()


def bits(*fs):
    m = 0
    for f in fs:
        m |= 1 << f
    return m


@pytest.fixture(scope='module')
def base_sd():
    return synth.make_backbone_state(seed=31)


def test_seeded_network_is_inside_the_fp16_window(lib, base_sd):
    u1, u16, bound, wmean, werr = verdict(lib, base_sd)
    assert u1 == 0 and u16 == 0
    for f in range(2, 19):
        assert 1.0 < bound[f] < 1000.0, (f, bound[f])          # interval bounds of a batch-normalised network: tens to hundreds
        assert wmean[f] > 1.0 and werr[f] < 0.1


def test_bounds_dominate_what_the_network_really_produces(lib, base_sd):
    """The overflow side is a proof only if the interval bound is a bound: compare with the oracle's activations."""
    from oracle import backbone_torch
    _, _, bound, _, _ = verdict(lib, base_sd)
    x = synth.normalize_crops(adv.extreme_crops(6))
    _, _, feats = backbone_torch.mobilenet_v2_forward(base_sd, x, return_features=True)
    last = {}
    for L in synth.mbv2_layers():
        last[L['feature']] = L['key']
    for f in range(1, 18):
        seen = float(feats[last[f]].abs().max())
        assert seen <= bound[f + 1] * (1 + 1e-5), (f, seen, bound[f + 1])
        assert bound[f + 1] < 256 * seen                        # and the slack the underflow estimate assumes (2^8) holds here


def test_huge_residual_stream_sends_its_blocks_to_fp32(lib, base_sd):
    u1, u16, bound, _, _ = verdict(lib, adv.scale_stream(base_sd, '64', 2.0 ** 14))
    assert u1 == bits(8, 9, 10, 11) and (u16 & u1) == u1
    assert bound[8] > 6.0e4
    u1, u16, _, _, _ = verdict(lib, adv.scale_stream(base_sd, '32', 2.0 ** 14))
    assert u1 == bits(5, 6, 7)


def test_medium_stream_only_leaves_the_times16_kernels(lib, base_sd):
    u1, u16, bound, _, _ = verdict(lib, adv.scale_stream(base_sd, '64', 2.0 ** 8))
    assert u1 == 0 and u16 == bits(8, 9, 10, 11)
    assert bound[11] < 6.0e4 < 16 * bound[8]


def test_tiny_residual_stream_is_flagged_as_underflow(lib, base_sd):
    u1, u16, bound, wmean, _ = verdict(lib, adv.scale_stream(base_sd, '160', 2.0 ** -20))
    assert u1 == bits(15, 16, 17) and wmean[15] < 0.25 / 16
    assert bound[15] < 1e-3


def test_rows_six_decades_apart(lib, base_sd):
    # with their BatchNorm shift the small rows' outputs are the shift: the split error of the weights is irrelevant -> fp16x2 stays
    u1, u16, _, _, werr = verdict(lib, adv.spread_rows(base_sd, 5, 6.0, zero_shift=False))
    assert u1 == 0 and werr[5] < 1.0
    # without it the rows' whole output is the GEMM result of weights that sit in fp16's subnormals -> exact kernel
    u1, u16, _, _, werr = verdict(lib, adv.spread_rows(base_sd, 5, 6.0, zero_shift=True))
    assert u1 == bits(5) and werr[5] > 1.0
    u1, _, _, _, _ = verdict(lib, adv.spread_rows(base_sd, 12, 6.0, zero_shift=True))
    assert u1 == bits(12)


def test_degenerate_checkpoints_do_not_break_the_packer(lib, base_sd):
    sd = {k: np.array(v, copy=True) for k, v in base_sd.items()}
    sd['features.9.conv.0.0.weight'][:] = 0                     # an all-zero layer: scale exponent 0, nothing to split
    u1, _, _, _, _ = verdict(lib, sd)
    assert u1 == 0
    sd['features.9.conv.0.0.weight'][:] = 1e-38                 # subnormal-range weights: the scale exponent is clamped, the block goes exact
    u1, _, _, _, _ = verdict(lib, sd)
    assert u1 & bits(9)
    sd['features.9.conv.0.0.weight'][0, 0] = np.nan
    u1, _, _, _, _ = verdict(lib, sd)
    assert u1 & bits(9)
