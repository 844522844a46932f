"""ctypes binding of the C ABI in include/synergy_hip.h (libsynergy_hip.so).

There is no CPU fallback: if the HIP library is missing or a call fails this raises.
torch is imported first on purpose so that the library's libamdhip64.so.7 dependency
resolves to the HIP runtime torch already loaded (one runtime per process; device
pointers and streams are shared with torch).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede loading the library, see module docstring)

from .build import LIB

SYN_ERR_PARAM_LEN = -4
_lib = None


class SynergyHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libsynergy_hip: {msg} (status {code})')
        self.code = code
        self.msg = msg


_SIGS = {
    'syn_last_error': (C.c_char_p, []),
    'syn_abi_version': (C.c_int, []),
    'syn_create': (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    'syn_destroy': (C.c_int, [C.c_void_p]),
    'syn_backbone_flat_count': (C.c_size_t, []),
    'syn_load_backbone': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'syn_numerics_report': (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    'syn_set_schedule': (C.c_int, [C.c_void_p, C.c_int]),
    'syn_backbone_range_status': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'syn_backbone_range_events': (C.c_int, [C.c_void_p]),
    'syn_backbone_calibrate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    'syn_resnet50_flat_count': (C.c_size_t, []),
    'syn_load_backbone_resnet50': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'syn_resnet50_flops_per_face': (C.c_double, []),
    'syn_load_basis': (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + [C.c_int, C.c_int]),
    'syn_constants_bytes': (C.c_size_t, [C.c_void_p]),
    'syn_export_constants': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'syn_import_constants': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'syn_bcast_constants': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'syn_describe_constants': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'syn_pack_constants_host_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'syn_pack_constants_host': (C.c_int, [C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    'syn_check_constants_host': (C.c_int, [C.c_void_p, C.c_size_t]),
    'syn_workspace_bytes': (C.c_size_t, [C.c_void_p, C.c_int]),
    'syn_backbone_forward': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_backbone_forward_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_crop_resize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    'syn_crop_resize_frames': (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.c_void_p]),
    'syn_detector_flat_count': (C.c_size_t, []),
    'syn_load_detector': (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    'syn_detector_prior_count': (C.c_int, [C.c_int, C.c_int]),
    'syn_detect': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                             C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    'syn_nme': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'syn_load_triangles': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    'syn_mesh_shade': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_rasterize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'syn_add_weighted': (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]),
    'syn_reconstruct': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_reconstruct_pitched': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'syn_pose': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_landmarks_pose': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_pose_matrix': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'syn_backbone_launch_count': (C.c_int, [C.c_void_p]),
    'syn_backbone_profile': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'syn_reconstruct_profile': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'syn_backbone_flops_per_face': (C.c_double, []),
    'syn_pointwise_flops_per_face': (C.c_double, []),
}

EXPORTED_SYMBOLS = tuple(_SIGS)          # exactly the symbols include/synergy_hip.h declares
# test hook exported by the library but deliberately not part of the public header
_SIGS = dict(_SIGS, syn_debug_feature=(C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
             syn_debug_profile_block=(C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
             syn_debug_detect_raw=(C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float] + [C.c_void_p] * 5),
             syn_debug_poison_workspace=(C.c_int, [C.c_void_p, C.c_int, C.c_int]))


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get('SYNERGY_HIP_LIB', LIB)      # A/B of two builds on one box (tools/ab_bench.sh); default: the in-tree build
        if not os.path.isfile(path):
            raise SynergyHipError(-100, f'{path} not built; run `python -c "import __graft_entry__ as g; g.build()"`')
        l = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().syn_last_error().decode(errors='replace')
        raise SynergyHipError(rc, msg)
