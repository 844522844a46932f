"""ORACLE (test infrastructure, NOT product code) -- authoring-container only.

Imports the REAL reference implementation (read-only checkout at
/root/reference) so that oracle/recon_numpy.py and oracle/backbone_torch.py can
be pinned against it and golden fixtures generated (tests/golden/make_golden.py).

The reference cannot be imported as-is here: torchvision / cv2 are
not installed, and 3dmm_data/ + pretrained/best.pth.tar do not exist.  Nothing is
copied: the reference package is imported from where it lies, with
  * stub modules for torchvision(.transforms), cv2 and FaceBoxes (none of them is
    touched below get_all_outputs' detector/resize lines, synergy3DMM.py:170-188),
  * utils.io._load patched to serve the synthetic 3DMM pack for the seven files
    utils/params.py:12-24 asks for,
  * scipy.io.loadmat patched for 3dmm_data/tri.mat (synergy3DMM.py:73).
/root/reference does not exist on the GPU box; callers must guard with available().
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REF_ROOT = '/root/reference'
_REF = None
_REF_MODULES = {}


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, 'synergy3DMM.py'))


def load_reference(pack: dict):
    """Returns the reference's `synergy3DMM` module, with ParamsPack fed from `pack`
    (dict from synergynet_amd.synth.make_3dmm)."""
    if not available():
        raise RuntimeError('reference checkout not present')
    global _REF
    if _REF is not None:
        return _REF
    saved = {n: sys.modules.pop(n) for n in list(sys.modules)
             if n.split('.')[0] in ('utils', 'loss_definition', 'synergy3DMM', 'backbone_nets', 'FaceBoxes')}

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if 'torchvision' not in sys.modules:
        tv = _stub('torchvision')
        tv.transforms = _stub('torchvision.transforms')
    if 'cv2' not in sys.modules:
        _stub('cv2')
    _stub('FaceBoxes', FaceBoxes=object)

    sys.path.insert(0, REF_ROOT)
    try:
        import scipy.io as sio
        import utils.io as ref_io           # the reference's utils package

        files = {
            'keypoints_sim.npy': pack['keypoints'], 'w_shp_sim.npy': pack['w_shp'],
            'w_exp_sim.npy': pack['w_exp'], 'u_shp.npy': pack['u_shp'], 'u_exp.npy': pack['u_exp'],
            'param_whitening.pkl': {'param_mean': pack['param_mean'], 'param_std': pack['param_std']},
        }
        orig_load = ref_io._load

        def _load(fp):
            base = os.path.basename(fp)
            if base in files:
                return files[base]
            return orig_load(fp)

        ref_io._load = _load
        orig_loadmat = sio.loadmat

        def _loadmat(fp, *a, **k):
            if os.path.basename(str(fp)) == 'tri.mat':
                return {'tri': pack['tri']}
            return orig_loadmat(fp, *a, **k)

        sio.loadmat = _loadmat               # stays patched: SynergyNet.__init__ calls it later (:73)
        import synergy3DMM as ref            # noqa: runs ParamsPack() three times
        _REF = ref
        return ref
    finally:
        sys.path.remove(REF_ROOT)
        # take the reference's modules back out of sys.modules so this repo's own
        # `synergy3DMM` shim / package names are not shadowed for the rest of the process
        for n in list(sys.modules):
            f = getattr(sys.modules[n], '__file__', None) or ''
            if f.startswith(REF_ROOT) or n == 'FaceBoxes':
                _REF_MODULES[n] = sys.modules.pop(n)
        sys.modules.update(saved)


def build_reference_model(pack: dict, backbone_sd: dict):
    """Real reference SynergyNet() with the synthetic checkpoint loaded through its own
    load_weights() (synergy3DMM.py:156-164; 'module.'-prefixed DataParallel keys)."""
    import tempfile

    import torch

    ref = load_reference(pack)
    model = ref.SynergyNet()                 # silent random init: best.pth.tar is absent (:109-113)
    ckpt = {'module.I2P.backbone.' + k: torch.from_numpy(np.asarray(v)) for k, v in backbone_sd.items()}
    with tempfile.NamedTemporaryFile(suffix='.pth.tar') as f:
        torch.save({'epoch': 0, 'state_dict': ckpt}, f.name)
        model.load_weights(f.name)
    model.eval()
    return ref, model
