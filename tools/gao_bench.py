"""Wall clock of the documented entry point (reference synergy3DMM.py:167-207) in a fresh process: frames + detections in, numpy landmarks /
meshes / poses out, everything from the frame upload to the mesh download inside.  One JSON object on stdout (bench.py: extra.get_all_outputs)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from synergynet_amd import synth  # noqa: E402
from synergynet_amd.synergy3DMM import SynergyNet  # noqa: E402


def main():
    model = SynergyNet(device='cuda:0', pack=synth.make_3dmm(), backbone_state=synth.make_backbone_state())
    sync = torch.cuda.synchronize
    fr = [synth.make_frame(720, 1080, seed=40 + i) for i in range(16)]
    rr = np.random.default_rng(7)

    def boxes():
        out = []
        for _ in range(8):
            side = float(rr.uniform(90, 380)); x0 = float(rr.uniform(-20, 1080 - side * 0.8)); y0 = float(rr.uniform(-20, 720 - side * 0.8))
            out.append([x0, y0, x0 + side, y0 + side * float(rr.uniform(0.9, 1.2)), 0.9])
        return out
    gao = {}
    for tag, nf, dense in (('1_frame_x_8_faces', 1, True), ('16_frames_x_8_faces', 16, True), ('16_frames_x_8_faces_lmk_pose_only', 16, False)):
        ts = []
        for it in range(12):
            rl = [boxes() for _ in range(nf)]
            t0 = time.perf_counter()
            if nf == 1 and dense:
                res = model.get_all_outputs(fr[0], rects=rl[0])
            else:
                res = model.get_all_outputs_batch(fr[:nf], rl, dense=dense)
            ts.append(time.perf_counter() - t0)
            del res
        t = float(np.median(ts[2:]))
        gao[tag] = dict(ms_per_call=round(t * 1e3, 4), frames_s=round(nf / t, 1), faces_s=round(8 * nf / t, 1),
                        wall_us_per_face=round(t / (8 * nf) * 1e6, 2))
    # host share of a 128-face call: everything but waiting for the device and the DMA (crop tables, staging, list building)
    rl = [boxes() for _ in range(16)]
    sync()
    t0 = time.perf_counter()
    model.get_all_outputs_batch(fr, rl)
    t_all = time.perf_counter() - t0
    lt = model.last_timing
    gao['host_us_per_face'] = round(lt['host_s'] / lt['faces'] * 1e6, 2)
    gao['device_and_dma_wait_us_per_face'] = round(lt['device_wait_s'] / lt['faces'] * 1e6, 2)
    gao['what'] = ('720x1080 uint8 frames + 8 given detections each -> crop/Lanczos resize on device -> MobileNetV2 -> 68 landmarks, 53215-vertex '
                   'mesh, pose per face -> page-locked host arrays (one DMA per output kind); wall clock per call, median of 10')
    gao['last_call_ms'] = round(t_all * 1e3, 4)
    print(json.dumps(gao))


if __name__ == '__main__':
    main()
