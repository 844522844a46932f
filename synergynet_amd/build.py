"""Builds the gfx950 shared library (C ABI in include/synergy_hip.h) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the authoring container.
Every source is compiled to its own object (in parallel, only when it or a header changed), then linked.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(PKG, '_obj')
LIB = os.path.join(PKG, 'libsynergy_hip.so')
SOURCES = ['synergy_abi.hip', 'backbone_kernels.hip', 'fused_block.hip', 'fused_block_f16.hip', 'fused_block_early.hip', 'fused_block_rm.hip', 'fused_block_lb.hip', 'fused_block_lb4.hip', 'stem_block1.hip', 'stem_rm.hip', 'head_kernel.hip', 'resnet_kernels.hip', 'preproc_kernels.hip', 'recon_kernels.hip', 'render_kernels.hip', 'detector_kernels.hip', 'eval_kernels.hip']
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
FLAGS = CFLAGS + ['-shared']          # (kept for tools that print the full command line)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(PKG, '..', 'include', 'synergy_hip.h')]


def _newer(dep_list, target) -> bool:
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in dep_list if os.path.isfile(d))


def _signature(extra_flags) -> str:
    """What a build was made FROM, beyond file dates: the source list and the flags (a change of either must rebuild)."""
    import hashlib
    return hashlib.sha256(repr((SOURCES, CFLAGS, tuple(extra_flags))).encode()).hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = '') -> str:
    """Default build: synergynet_amd/libsynergy_hip.so from objects in _obj/.  A build with `extra_flags` is a VARIANT (A/B kernels,
    tools/build_variant.sh): it needs a `variant` name and lives entirely apart -- objects in _obj/<variant>/, library
    libsynergy_hip_<variant>.so (load it through SYNERGY_HIP_LIB) -- so it can never be mistaken for, or silently reused as, the
    default library."""
    if extra_flags and not variant:
        raise ValueError('build_library: extra_flags need a variant name (objects and library are kept apart from the default build)')
    obj = os.path.join(OBJ, variant) if variant else OBJ
    lib = LIB.replace('.so', f'_{variant}.so') if variant else LIB
    sig_file = os.path.join(obj, 'build.sig')
    sig = _signature(extra_flags)
    sig_ok = os.path.isfile(sig_file) and open(sig_file).read().strip() == sig
    force = force or not sig_ok
    if not force and not _newer([os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers(), lib):
        return lib
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc):
        raise RuntimeError('hipcc not found: cannot build libsynergy_hip.so')
    os.makedirs(obj, exist_ok=True)
    hdrs = _headers()

    def compile_one(src):
        sp, op = os.path.join(CSRC, src), os.path.join(obj, src.replace('.hip', '.o'))
        if not force and not _newer([sp] + hdrs, op):
            return None
        cmd = [hipcc] + CFLAGS + list(extra_flags) + ['-c', sp, '-o', op]
        if verbose:
            print(' '.join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        return (src, r.stdout + r.stderr) if r.returncode != 0 else None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        errs = [e for e in ex.map(compile_one, SOURCES) if e]
    if errs:
        raise RuntimeError('hipcc failed:\n' + '\n'.join(f'--- {s}\n{msg}' for s, msg in errs))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib + '.tmp'] + [os.path.join(obj, s.replace('.hip', '.o')) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc link failed:\n' + r.stdout + r.stderr)
    os.replace(lib + '.tmp', lib)
    with open(sig_file, 'w') as f:
        f.write(sig + '\n')
    return lib


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
