"""CPU-side tests (no GPU): the C-ABI library builds/loads and exports every symbol the public
header declares, and the host logic around it (key flattening, ParamsPack, crop/resize)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def test_library_builds_and_exports_header_symbols():
    from synergynet_amd.build import build_library
    from synergynet_amd import abi
    lib = build_library()
    assert os.path.isfile(lib)
    hdr = open(os.path.join(ROOT, 'include', 'synergy_hip.h')).read()
    declared = set(re.findall(r'\b(syn_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    import torch  # noqa: F401  -- the library must bind to torch's HIP runtime
    l = ctypes.CDLL(lib)
    for s in declared:
        assert hasattr(l, s), s
    l.syn_abi_version.restype = ctypes.c_int
    assert l.syn_abi_version() == 1
    l.syn_backbone_flat_count.restype = ctypes.c_size_t
    l.syn_backbone_flops_per_face.restype = ctypes.c_double
    l.syn_pointwise_flops_per_face.restype = ctypes.c_double
    # 93,204,560 MAC per face; pointwise share 83.81 M MAC
    assert abs(l.syn_backbone_flops_per_face() - 2 * 93204560) < 1
    assert abs(l.syn_pointwise_flops_per_face() / 2e6 - 83.81) < 0.01


def test_flat_backbone_layout_matches_library(backbone_sd):
    from synergynet_amd.synergy3DMM import backbone_keys, flatten_backbone
    from synergynet_amd.build import build_library
    import torch  # noqa: F401
    keys = backbone_keys()
    assert keys[0][0] == 'features.0.0.weight' and keys[-1][0] == 'classifier_exp.1.bias'
    n_conv = sum(1 for k, _ in keys if k.endswith('.weight') and len(_) == 4)
    assert n_conv == 52                                  # 52 convs
    flat = flatten_backbone(backbone_sd)
    l = ctypes.CDLL(build_library())
    l.syn_backbone_flat_count.restype = ctypes.c_size_t
    assert flat.size == l.syn_backbone_flat_count()
    bad = dict(backbone_sd)
    bad['features.3.conv.2.weight'] = np.zeros((24, 100, 1, 1), np.float32)
    with pytest.raises(RuntimeError, match='shape'):
        flatten_backbone(bad)


def test_resnet50_flat_layout_matches_library():
    from synergynet_amd import synth
    from synergynet_amd.synergy3DMM import flatten_backbone, resnet50_keys
    from synergynet_amd.build import build_library
    import torch  # noqa: F401
    sd = synth.make_resnet50_state()
    keys = resnet50_keys()
    assert sum(1 for k, sh in keys if len(sh) == 4) == 53          # 1 stem + 16*3 + 4 downsample convs
    flat = flatten_backbone(sd, arch='resnet50')
    l = ctypes.CDLL(build_library())
    l.syn_resnet50_flat_count.restype = ctypes.c_size_t
    l.syn_resnet50_flops_per_face.restype = ctypes.c_double
    assert flat.size == l.syn_resnet50_flat_count()
    # ResNet-50 at 120x120 = 1.259 G MAC = 2.518 GFLOP per face
    assert abs(l.syn_resnet50_flops_per_face() / 2.518e9 - 1) < 0.01


def test_state_dict_keys_equal_reference_keys(backbone_sd):
    """The key tree must be loadable from a reference checkpoint (synergy3DMM.py:156-164)."""
    from synergynet_amd.synergy3DMM import backbone_keys
    ours = {k for k, _ in backbone_keys()}
    theirs = {k for k in backbone_sd if not k.endswith('num_batches_tracked')}
    assert ours == theirs
    from oracle import ref_loader
    if ref_loader.available():
        import sys
        sys.path.insert(0, ref_loader.REF_ROOT)
        try:
            import importlib
            m = importlib.import_module('backbone_nets.mobilenetv2_backbone')
            ref_keys = {k for k in m.mobilenet_v2().state_dict() if not k.endswith('num_batches_tracked')}
        finally:
            sys.path.remove(ref_loader.REF_ROOT)
            for n in [n for n in sys.modules if n.startswith('backbone_nets')]:
                del sys.modules[n]
        assert ours == ref_keys


def test_params_pack_mirror(pack, tmp_path):
    from synergynet_amd.params import ParamsPack
    pp = ParamsPack(pack=pack)
    assert pp.std_size == 120 and pp.dim == 53215
    assert pp.u.shape == (159645, 1) and pp.u_base.shape == (204, 1)
    assert pp.w_shp_base.shape == (204, 40) and pp.w_exp_base.shape == (204, 10)
    assert np.array_equal(pp.u_base[:, 0], (pack['u_shp'] + pack['u_exp'])[pack['keypoints'], 0])
    with pytest.raises(RuntimeError, match='Missing data'):
        ParamsPack(data_dir=str(tmp_path))
    # file-based path (utils/params.py:12-24) with a small pack
    import pickle
    from synergynet_amd import synth
    small = synth.make_3dmm(seed=1, n_vert=200)
    d = tmp_path / '3dmm_data'
    d.mkdir()
    np.save(d / 'keypoints_sim.npy', small['keypoints'])
    np.save(d / 'w_shp_sim.npy', small['w_shp'])
    np.save(d / 'w_exp_sim.npy', small['w_exp'])
    np.save(d / 'u_shp.npy', small['u_shp'])
    np.save(d / 'u_exp.npy', small['u_exp'])
    with open(d / 'param_whitening.pkl', 'wb') as f:
        pickle.dump({'param_mean': small['param_mean'], 'param_std': small['param_std']}, f)
    pf = ParamsPack(data_dir=str(d))
    assert pf.dim == 200 and np.array_equal(pf.w_shp, small['w_shp'])


def test_crop_img_zero_pads_like_reference():
    from oracle.preproc_numpy import crop_img
    img = np.arange(10 * 12 * 3, dtype=np.uint8).reshape(10, 12, 3)
    c = crop_img(img, [-2.4, -1.6, 5.2, 4.4, 1.0])          # rounds to [-2,-2,5,4]
    assert c.shape == (6, 7, 3)
    assert (c[:2] == 0).all() and (c[:, :2] == 0).all()
    assert np.array_equal(c[2:, 2:], img[0:4, 0:5])
    c = crop_img(img, [8, 7, 15, 13, 1.0])                  # overhang right/bottom
    assert c.shape == (6, 7, 3) and np.array_equal(c[:3, :4], img[7:10, 8:12]) and (c[3:] == 0).all()


def test_resize_lanczos4_basic_properties():
    from oracle.preproc_numpy import resize_lanczos4
    flat = np.full((200, 170, 3), 97, np.uint8)
    assert (resize_lanczos4(flat, 120, 120) == 97).all()     # partition of unity
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (120, 120, 3), dtype=np.uint8)
    assert np.array_equal(resize_lanczos4(img, 120, 120), img)   # identity at scale 1
    ramp = np.tile(np.linspace(0, 255, 240).astype(np.uint8)[None, :, None], (240, 1, 3))
    r = resize_lanczos4(ramp, 120, 120)
    assert r.shape == (120, 120, 3) and (np.diff(r[60, :, 0].astype(int)) >= -1).all()


def test_synthetic_assets_are_deterministic():
    from synergynet_amd import synth
    a, b = synth.make_backbone_state(5), synth.make_backbone_state(5)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    p, q = synth.make_3dmm(9, n_vert=300), synth.make_3dmm(9, n_vert=300)
    assert all(np.array_equal(p[k], q[k]) for k in p)
    assert len(set(int(v) // 3 for v in p['keypoints'])) == 68


def test_documented_knobs_exist_in_the_sources():
    """Every environment knob INTEGRATION.md / DESIGN.md name (SYN_* / SYNERGY_HIP_*) is read somewhere in the library, the bench or the tools --
    a renamed knob must not survive in the documentation."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = ''
    for pat in ('synergynet_amd/csrc/*', 'synergynet_amd/*.py', 'bench.py', 'tools/*', '__graft_entry__.py', 'tests/*.py'):
        for f in glob.glob(os.path.join(root, pat)):
            if os.path.isfile(f) and not f.endswith(('.so', '.o')):
                try:
                    src += open(f, errors='replace').read()
                except OSError:
                    pass
    missing = []
    for doc in ('INTEGRATION.md', 'DESIGN.md'):
        text = open(os.path.join(root, doc)).read()
        for knob in sorted(set(re.findall(r'\b(SYN(?:ERGY_HIP)?_[A-Z0-9_]*[A-Z0-9])\b', text))):
            stem = re.sub(r'_?<[^>]*>.*$', '', knob)
            if knob.endswith('_MIN') and knob + '<' in text:          # SYN_LB_MIN<f>: the feature number is appended at run time
                stem = knob
            if stem not in src:
                missing.append(f'{doc}: {knob}')
    # ... and every name INTEGRATION.md lists for SYNERGY_HIP_TEST_KNOBS is a test_knob("...") somewhere in csrc/
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    m = re.search(r'SYNERGY_HIP_TEST_KNOBS.*?[Nn]ames?:(.*?)\n\n', text, re.S)
    assert m, 'INTEGRATION.md documents SYNERGY_HIP_TEST_KNOBS and its names'
    for name in re.findall(r'`([a-z0-9_]+)`', m.group(1)):
        if f'test_knob("{name}"' not in src and f'test_knob_set("{name}"' not in src:
            missing.append(f'INTEGRATION.md: test knob {name}')
    assert not missing, missing


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r4/bench_b1024.json is a bench.py line from the GPU box: the fields the driver and the judge read must be there and consistent
    (whole-job value = faces per step / step time, roofline fraction = achieved / peak, a bounded CPU baseline with its core count)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, 'profiles', 'r4', 'bench_b1024.json')))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    B = d['config']['global_batch']
    assert abs(d['value'] - B / d['ms_per_step'] * 1e3) / d['value'] < 1e-3
    r = d['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] in ('GB/s', 'TFLOP/s')
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] < 1
    c = d['cpu_baseline']
    assert c['kind'] in ('reference', 'port') and c['cores'] >= 1 and c['value'] > 0 and c['sample']
    # round 3: the arithmetic is named for what it is, the counter-derived fields say where they come from, the guard's verdict is in the line
    assert d['dtype'].startswith('f32') and 'fp16x2' in d['dtype']
    src = r['counters_source']
    assert src['measured_by_this_run'] is False and src['file'].startswith('profiles/traffic_r') and src['commit'] and src['box']
    assert os.path.isfile(os.path.join(root, src['file']))
    assert d['numerics']['fallback_blocks'] == 0
    g = d['extra']['get_all_outputs']
    assert g['host_us_per_face'] < 50 and g['16_frames_x_8_faces']['faces_s'] > 1e4
    k = d['extra']['reconstruction_alone']['kernel']
    assert k['bound'] == 'hbm' and abs(k['frac'] - k['achieved'] / k['peak']) < 1e-3 and 0 < k['frac'] < 1


def test_roofline_fraction_can_be_recomputed_from_the_committed_kernel_stats():
    """frac = algorithmic family FLOPs / sum of the family kernels' time per forward / ceiling, recomputed from
    profiles/r4/kernel_stats_b1024_one_stream.csv (rocprofv3 --kernel-trace --stats of the same command), must agree with the printed
    roofline.frac within 8 % (the profiler's own slowdown is ~3-7 %)."""
    import csv
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, 'profiles', 'r4', 'bench_b1024.json')))
    rows = list(csv.DictReader(open(os.path.join(root, 'profiles', 'r4', 'kernel_stats_b1024_one_stream.csv'))))
    forwards = [int(r['Calls']) for r in rows if 'stem_rm_kernel' in r['Name']][0]
    fam_ns = sum(float(r['TotalDurationNs']) for r in rows if 'fused_block' in r['Name'] or 'fused_chain' in r['Name']) / forwards
    r = d['roofline']
    n_launch = len([p for p in r['per_launch'] if 2 <= p['feature'] <= 17 or p['feature'] >= 100])
    fam_flops = r['flops_per_launch'] * n_launch
    frac = fam_flops / (fam_ns * 1e-9) / 1e12 / r['peak']
    assert abs(frac - r['frac']) / r['frac'] < 0.08, (frac, r['frac'])
    # and the profile bundle is the one the line says its counters are from
    t = json.load(open(os.path.join(root, r['counters_source']['file'])))
    assert t['commit'] == r['counters_source']['commit']


def test_face_tables_equal_the_per_face_arithmetic():
    """get_all_outputs_batch's array form of the ROI / crop-box / tap-table preparation (synergy3DMM.py _face_tables) against the
    scalar statement of reference synergy3DMM.py:178-185 + utils/inference.py:98 (_roi_and_box): same bits, same in-place mutation of
    the detection lists; halves (x.5 box edges, margins on the floor-division boundary) included."""
    import copy
    from synergynet_amd.inference import lanczos4_tables
    from synergynet_amd.synergy3DMM import SynergyNet
    rng = np.random.default_rng(5)
    rects = []
    for f in range(6):
        fr = []
        for i in range(f):                                   # frame 0 has no face
            x, y = rng.uniform(-40, 400), rng.uniform(-40, 300)
            s = [rng.uniform(30, 220), 100.0, 83.5, 125.0 / 1.2, 64.0][i % 5]
            fr.append([x if i % 2 else float(round(x)) + 0.5, y, x + s, y + s * rng.uniform(0.9, 1.2), float(rng.uniform(0.5, 1))])
        rects.append(fr)
    n = sum(len(fr) for fr in rects)
    want_rects = copy.deepcopy(rects)
    roi = np.empty((n, 5), np.float32); box = np.empty((n, 4), np.int32)
    ofs = np.empty((2, n, 120), np.int32); coef = np.empty((2, n, 120, 8), np.int16)
    k = 0
    for fr in want_rects:
        for rect in fr:
            r5, b4 = SynergyNet._roi_and_box(None, rect)
            roi[k], box[k] = r5, b4
            ofs[0, k], coef[0, k] = lanczos4_tables(b4[2] - b4[0])
            ofs[1, k], coef[1, k] = lanczos4_tables(b4[3] - b4[1])
            k += 1
    got = SynergyNet._face_tables(rects, n)
    for g, w in zip(got, (roi, box, ofs, coef)):
        assert g.dtype == w.dtype and np.array_equal(g, w)
    assert rects == want_rects
    with pytest.raises(ValueError):
        SynergyNet._face_tables([[[10.0, 10.0, 20.0, 10.2, 0.9]]], 1)
    # detections as numpy float32 scalars (what FaceBoxes returns, here and in the reference): the scalar statement then computes in
    # float32 (the array form in double moved the ROI by an ulp in a third of the cases, a crop box by a pixel in 1 of 10^4);
    # and mixed element types go through the scalar statement itself
    for mode in ('f32', 'mixed'):
        rng = np.random.default_rng(11)
        r32 = []
        for i in range(4000):
            x, y, s_ = rng.uniform(-40, 400), rng.uniform(-40, 300), rng.uniform(20, 260)
            v = [np.float32(x), np.float32(y), np.float32(x + s_), np.float32(y + s_ * rng.uniform(0.9, 1.2)), np.float32(rng.uniform(0.5, 1))]
            if mode == 'mixed' and i % 3 == 0:
                v[2] = float(v[2])
            r32.append(v)
        want32 = copy.deepcopy(r32)
        wroi = np.empty((len(r32), 5), np.float32); wbox = np.empty((len(r32), 4), np.int32)
        for k, rect in enumerate(want32):
            wroi[k], wbox[k] = SynergyNet._roi_and_box(None, rect)
        groi, gbox, _, _ = SynergyNet._face_tables([r32], len(r32))
        assert np.array_equal(groi, wroi) and np.array_equal(gbox, wbox), mode
        assert all(type(a) is type(b) and a == b for ra, rb in zip(r32, want32) for a, b in zip(ra, rb)), mode


def test_frame_chunks_partition_the_call():
    """get_all_outputs_batch's chunking (synergy3DMM.py _chunks): consecutive whole frames, every face exactly once and in order, at least
    chunk_faces faces per chunk whenever there is more than one chunk, frames without faces anywhere."""
    from synergynet_amd.synergy3DMM import SynergyNet
    rng = np.random.default_rng(9)
    cases = [[8] * 16, [0, 0, 5], [3], [0, 7, 0, 0, 9, 1, 0], [1] * 40, [64, 64], [63, 64], [200, 1, 1]]
    cases += [list(rng.integers(0, 12, size=int(rng.integers(1, 30)))) for _ in range(50)]
    for counts in cases:
        n = int(sum(counts))
        if n == 0:
            continue
        for cf in (1, 2, 5, 32, 64, 1000, 0):
            ch = SynergyNet._chunks(counts, cf)
            assert ch[0][0] == 0 and ch[0][2] == 0 and ch[-1][1] == len(counts) and ch[-1][3] == n
            for (a0, a1, l0, l1), (b0, b1, m0, m1) in zip(ch, ch[1:]):
                assert a1 == b0 and l1 == m0
            for f0, f1, l0, l1 in ch:
                assert f0 < f1 and l1 - l0 == sum(counts[f0:f1])
                assert len(ch) == 1 or l1 - l0 >= max(1, cf)
            if n < 2 * max(1, cf):
                assert len(ch) == 1


def test_hot_loops_never_drain_their_loads_in_flight():
    """ISA regression guard for round 3's main finding: in the steady-state loops of the LDS-tiled ResNet GEMM and of the
    MobileNetV2 tail no `s_waitcnt vmcnt(0)` may appear and no wait may sit directly behind a load -- a select behind a buffer load, a
    conditional fetch inside the loop, or a loop header whose entry state differs from its back edge each made the compiler drain every
    load in flight once per step (92 -> 55 us for layer 3's conv1).  Cross-compiles on the CPU (hipcc -S), no GPU."""
    import shutil
    import sys
    if not (shutil.which('hipcc') or os.path.isfile('/opt/rocm/bin/hipcc')):
        pytest.skip('no hipcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    try:
        import isa_scan
    finally:
        sys.path.pop(0)
    # kernels are matched by PREFIX of the demangled name (a new trailing template parameter must not blind the guard: round 5 ended red
    # on exactly that), and "kernel not found" is reported apart from "loop drains"
    checks = [('resnet_kernels.hip', ['conv_lt_kernel<4', 'conv_lt_kernel<2'], 48), ('head_kernel.hip', ['head_f16x2_kernel<1, 4, 8'], 100)]
    for src, kernels, min_mfma in checks:
        loops = isa_scan.loop_sequences(isa_scan.compile_to_asm(src), kernels)
        for k in kernels:
            assert any(name.startswith('syn::' + k) or name.startswith('void syn::' + k) or k in name for name in loops), \
                f'{src}: no kernel whose name starts with {k!r} in the ISA (renamed? the guard must follow it): found {sorted(loops)}'
        for name, ls in loops.items():
            main = [l for l in ls if l[2] >= min_mfma]
            assert main, f'{name}: no steady-state loop found'
            for a, b, nm, seq in main:
                assert 'w0' not in seq, f'{name}: vmcnt(0) inside the loop at lines {a}-{b}: {" ".join(seq)}'
                for x, y in zip(seq, seq[1:]):
                    assert not (x == 'L' and y.startswith('w') and int(y[1:]) <= 1), f'{name}: a wait right behind a load: {" ".join(seq)}'
                assert any(t.startswith('w') and int(t[1:]) >= 6 for t in seq), f'{name}: no counted wait (loads in flight across steps) left: {" ".join(seq)}'


def test_write_obj_files_are_byte_identical_to_the_reference(tmp_path):
    """`write_obj` (reference utils/inference.py:8-23).  Golden = the bytes the reference's OWN function wrote for
    a seeded mesh (tests/golden/make_golden.py main_write_obj): values that round at the fourth decimal, negative zero, 1e5."""
    from synergynet_amd.inference import write_obj
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'write_obj_golden.npz'))
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        write_obj('mesh', g['vertices'], g['triangles'])                 # '.obj' is appended
        write_obj('mesh2.obj', g['vertices'], g['triangles'])            # ... only where it is missing
        write_obj('empty', g['vertices'][:, :0], g['triangles'][:, :0])
    finally:
        os.chdir(cwd)
    assert str(g['name_plain']) == 'mesh.obj' and open(tmp_path / 'mesh.obj', 'rb').read() == g['bytes_plain'].tobytes()
    assert open(tmp_path / 'mesh2.obj', 'rb').read() == g['bytes_with_ext'].tobytes()
    assert open(tmp_path / 'empty.obj', 'rb').read() == b''
    # float-typed triangle indices print their repr in the reference ('{}'.format): kept
    write_obj(str(tmp_path / 'f'), g['vertices'][:, :2], np.array([[1.0], [2.0], [1.0]]))
    assert open(tmp_path / 'f.obj').read().splitlines()[-1] == 'f 1.0 2.0 1.0'


def test_bcast_constants_rejects_null_arguments_without_a_device():
    """syn_bcast_constants exists in the C ABI and refuses NULL arguments before touching RCCL or HIP."""
    from synergynet_amd.build import build_library
    lib = ctypes.CDLL(build_library())
    lib.syn_last_error.restype = ctypes.c_char_p
    lib.syn_bcast_constants.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    assert lib.syn_bcast_constants(None, None, 0, None) == -1
    assert b'syn_bcast_constants' in lib.syn_last_error()
    hdr = ctypes.create_string_buffer(256)
    lib.syn_describe_constants.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    assert lib.syn_describe_constants(None, hdr, 256) == -1
