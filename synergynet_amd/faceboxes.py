"""FaceBoxes face detector on the GPU behind the reference's class surface.

    from FaceBoxes import FaceBoxes            # root shim -> this module
    face_boxes = FaceBoxes()                   # loads FaceBoxes/weights/FaceBoxesProd.pth like the reference (FaceBoxes.py:29,50)
    rects = face_boxes(img_bgr_uint8)          # [[xmin, ymin, xmax, ymax, score], ...] with score > 0.5 (FaceBoxes.py:60-143)

Everything from the uint8 frame to the NMS result runs in HIP kernels (csrc/detector_kernels.hip) through `syn_detect`; the
host only computes the down-scaling factor and applies the visualisation threshold, as the reference does in Python.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import abi, synth

confidence_threshold, top_k, keep_top_k, nms_threshold, vis_thres = 0.05, 5000, 750, 0.3, 0.5     # FaceBoxes.py:18-22
scale_flag, HEIGHT, WIDTH = True, 720, 1080                                                        # FaceBoxes.py:25-26


def flatten_detector(sd) -> np.ndarray:
    """state_dict (numpy arrays or tensors, optional 'module.' prefixes as load_model strips them, utils/functions.py:21-25)
    -> the flat float32 vector syn_load_detector takes (order documented in include/synergy_hip.h)."""
    sd = {(k.split('module.', 1)[-1] if k.startswith('module.') else k): v for k, v in sd.items()}
    get = lambda k: np.asarray(sd[k].detach().cpu().numpy() if isinstance(sd[k], torch.Tensor) else sd[k], dtype=np.float32).reshape(-1)
    parts = []
    for name, cin, cout, k, _, _, kind in synth.faceboxes_convs():
        if kind == 'head':
            parts += [get(name + '.weight'), get(name + '.bias')]
        else:
            parts += [get(name + '.conv.weight'), get(name + '.bn.weight'), get(name + '.bn.bias'), get(name + '.bn.running_mean'),
                      get(name + '.bn.running_var')]
    flat = np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)
    return flat


class FaceBoxes:
    """reference FaceBoxes/FaceBoxes.py:46-143."""

    def __init__(self, timer_flag=False, state_dict=None, weights_path=None, device='cuda:0'):
        if not torch.cuda.is_available():
            raise RuntimeError('FaceBoxes needs an MI355X GPU (there is no CPU path)')
        self.device = torch.device(device)
        self.timer_flag = timer_flag
        self._lib = abi.lib()
        self._h = C.c_void_p()
        abi.check(self._lib.syn_create(self.device.index or 0, C.byref(self._h)))
        if state_dict is None:
            path = weights_path or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'FaceBoxes', 'weights',
                                                'FaceBoxesProd.pth')
            if not os.path.isfile(path):
                # the reference prints this and exits the interpreter (utils/functions.py:30-32); raising is kinder to callers
                raise RuntimeError(f'The pre-trained FaceBoxes model {path} does not exist')
            ck = torch.load(path, map_location='cpu')
            state_dict = ck['state_dict'] if 'state_dict' in ck else ck
        flat = flatten_detector(state_dict)
        if flat.size != self._lib.syn_detector_flat_count():
            raise RuntimeError(f'FaceBoxes state_dict has {flat.size} values, expected {self._lib.syn_detector_flat_count()}')
        abi.check(self._lib.syn_load_detector(self._h, flat.ctypes.data_as(C.c_void_p), flat.size))
        self._dets = torch.empty((keep_top_k, 5), dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, '_h', None):
                self._lib.syn_destroy(self._h)
        except Exception:
            pass

    @staticmethod
    def frame_scale(h, w):
        """FaceBoxes.py:63-70."""
        scale = 1
        if scale_flag:
            if h > HEIGHT:
                scale = HEIGHT / h
            if w * scale > WIDTH:
                scale *= WIDTH / (w * scale)
        return scale

    @staticmethod
    def scaled_size(h, w, scale):
        """FaceBoxes.py:71-75: the network input size, int() of a python-double product exactly like the reference (a float32
        product differs by one pixel on ~8 % of frame sizes, which would move the resize taps and the prior grid)."""
        if scale == 1:
            return h, w
        return int(scale * h), int(scale * w)

    def detect_all(self, img_):
        """Rows before the vis_thres filter: float32 [n,5] (x1, y1, x2, y2, score), score-descending.  img_: uint8 [H,W,3] BGR,
        numpy array or device tensor."""
        frame = img_ if isinstance(img_, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_))
        if frame.dtype != torch.uint8 or frame.dim() != 3 or frame.shape[2] != 3:
            raise ValueError('frame must be uint8 [H,W,3] (BGR)')
        frame = frame.to(self.device).contiguous()
        h, w = int(frame.shape[0]), int(frame.shape[1])
        scale = self.frame_scale(h, w)
        h_s, w_s = self.scaled_size(h, w, scale)
        n = C.c_int(0)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            abi.check(self._lib.syn_detect(self._h, frame.data_ptr(), h, w, h_s, w_s, float(scale), confidence_threshold, nms_threshold,
                                           top_k, keep_top_k, self._dets.data_ptr(), C.byref(n), stream))
        return self._dets[:n.value].cpu().numpy()

    def __call__(self, img_):
        dets = self.detect_all(img_)
        return [[b[0], b[1], b[2], b[3], b[4]] for b in dets if b[4] > vis_thres]      # FaceBoxes.py:131-141
