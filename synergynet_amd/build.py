"""Builds the gfx950 shared library (C ABI in include/synergy_hip.h) in-tree with hipcc.

The .so is git-ignored but travels to the GPU box with the repo snapshot.  hipcc
cross-compiles for gfx950 without a GPU, so this also runs in the authoring container.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libsynergy_hip.so')
SOURCES = ['synergy_abi.hip', 'backbone_kernels.hip', 'fused_block.hip', 'fused_block_bf3.hip', 'fused_block_early.hip', 'stem_block1.hip', 'head_kernel.hip', 'resnet_kernels.hip', 'preproc_kernels.hip', 'recon_kernels.hip', 'render_kernels.hip', 'detector_kernels.hip', 'eval_kernels.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


def _stale() -> bool:
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, '..', 'include', 'synergy_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.isfile(hipcc):
        raise RuntimeError('hipcc not found: cannot build libsynergy_hip.so')
    cmd = [hipcc] + FLAGS + ['-o', LIB + '.tmp'] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + r.stdout + r.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build_library(force=True, verbose=True))
