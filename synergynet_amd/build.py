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
SOURCES = ['synergy_abi.hip', 'backbone_kernels.hip', 'fused_block.hip', 'fused_block_bf3.hip', 'fused_block_early.hip', 'fused_block_rm.hip', 'fused_block_lb.hip', 'fused_block_lb4.hip', 'stem_block1.hip', 'stem_rm.hip', 'head_kernel.hip', 'resnet_kernels.hip', 'preproc_kernels.hip', 'recon_kernels.hip', 'render_kernels.hip', 'detector_kernels.hip', 'eval_kernels.hip']
CFLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
FLAGS = CFLAGS + ['-shared']          # (kept for tools that print the full command line)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(PKG, '..', 'include', 'synergy_hip.h')]


def _newer(dep_list, target) -> bool:
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in dep_list if os.path.isfile(d))


def _stale() -> bool:
    return _newer([os.path.join(CSRC, f) for f in os.listdir(CSRC)] + _headers(), LIB)


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    if not force and not extra_flags and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc):
        raise RuntimeError('hipcc not found: cannot build libsynergy_hip.so')
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _headers()

    def compile_one(src):
        sp, op = os.path.join(CSRC, src), os.path.join(OBJ, src.replace('.hip', '.o'))
        if not force and not extra_flags and not _newer([sp] + hdrs, op):
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
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc link failed:\n' + r.stdout + r.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
