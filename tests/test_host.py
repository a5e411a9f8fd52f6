"""Host logic on CPU: JSON helpers vs the reference's goldens, IoU matching, fp16 splitting, BN folding,
w3*w2 merging and line-format packing (through the C ABI's host-only mode -- no GPU involved)."""
import copy
import ctypes
import json
import os

import numpy as np
import pytest
import torch

import synth
from monoloco_amd import _lib
from monoloco_amd.network import process as P
from monoloco_amd.utils import iou as I

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def c1():
    return json.load(open(os.path.join(G, 'golden_c1.json')))


@pytest.fixture(scope='module')
def ann():
    return json.load(open(os.path.join(G, 'pifpaf_002282.json')))


@pytest.mark.parametrize("name,kwargs", [('predict', dict(im_size=(1238, 374), enlarge_boxes=False)),
                                         ('default', dict(im_size=None)), ('eval', dict(im_size=(1242, 374))),
                                         ('minconf', dict(im_size=(1238, 374), min_conf=0.55))])
def test_preprocess_pifpaf_matches_reference(c1, ann, name, kwargs):
    a = copy.deepcopy(ann)
    boxes, kps = P.preprocess_pifpaf(a, **kwargs)
    assert boxes == c1['pre_' + name]['boxes']
    assert kps == c1['pre_' + name]['keypoints']
    # side effect of the reference: the caller's bbox lists are edited in place and get conf appended
    kept = [x for x in a if len(x['bbox']) == 5]
    assert len(kept) == len(boxes) and all(x['bbox'] is b for x, b in zip(kept, boxes))


def test_preprocess_pifpaf_score_branch(c1):
    boxes, kps = P.preprocess_pifpaf(copy.deepcopy(c1['pre_score']['input']), im_size=(1238, 374))
    assert boxes == c1['pre_score']['boxes'] and kps == c1['pre_score']['keypoints']


def test_prepare_pif_kps():
    assert P.prepare_pif_kps(list(range(9))) == [[0, 3, 6], [1, 4, 7], [2, 5, 8]]
    with pytest.raises(AssertionError):
        P.prepare_pif_kps([1, 2])


def test_load_calibration(c1):
    assert P.load_calibration('kitti', (1238, 374)) == c1['calib']['kitti_1238_374']
    assert P.load_calibration('kitti', (1242, 375)) == c1['calib']['kitti_1242_375']
    assert P.load_calibration('custom', (1920, 1080), focal_length=5.7) == c1['calib']['custom_1920_1080']
    assert P.load_calibration('nuscenes', (1600, 900)) == c1['calib']['nuscenes_1600_900']


def test_factory_for_gt(tmp_path):
    path = tmp_path / 'names.json'
    dic = {'000001.png': {'boxes': [[1, 2, 3, 4]], 'ys': [[0, 0, 0, 9.]], 'K': synth.KITTI_K}}
    path.write_text(json.dumps(dic))
    gt, kk = P.factory_for_gt(str(path), '000001.png')
    assert gt == dic['000001.png'] and kk == synth.KITTI_K
    with pytest.raises(AssertionError):
        P.factory_for_gt(str(tmp_path / 'missing.json'), 'x')


def test_iou_helpers():
    boxes = [[0, 0, 10, 10, 0.9], [20, 0, 30, 10, 0.5], [100, 100, 120, 130, 0.7], [1, 1, 11, 11, 0.95]]
    gts = [[0, 0, 10, 10], [21, 0, 31, 10], [300, 300, 310, 310]]
    assert I.calculate_iou(boxes[0], gts[0]) == 1.0
    assert I.calculate_iou(boxes[0], gts[2]) == 0.0
    assert I.get_iou_matrix(boxes, gts).shape == (4, 3)
    # highest confidence first: box 3 grabs gt 0, box 0's best gt is then taken -> unmatched
    assert I.get_iou_matches(boxes, gts, 0.3) == [(3, 0), (1, 1)]
    assert I.get_iou_matches([], gts) == [] and I.get_iou_matches(boxes, []) == []
    mm = I.get_iou_matches_matrix(boxes, gts, 0.3)
    assert [(int(a), int(b)) for a, b in mm] == [(0, 0), (1, 1)]
    assert I.reorder_matches([(1, 1), (3, 0)], boxes, mode='left_right') == [(3, 0), (1, 1)]


def test_extract_slices_and_labels():
    raw = torch.arange(20.).view(2, 10)
    assert [t.shape[1] for t in P.extract_outputs(raw, tasks=('d', 'x', 'ori', 'aux'))] == [2, 1, 2, 1]
    lab = torch.arange(22.).view(2, 11)
    assert P.extract_labels(lab, tasks=('d', 'aux'))[1][0, 0] == 10
    assert P.extract_labels_aux(lab)['aux'].shape == (2, 1)
    clustered = P.cluster_outputs(raw.repeat(3, 1), 3)
    assert clustered.shape == (2, 3, 10)
    with pytest.raises(AssertionError):
        P.cluster_outputs(raw.repeat(3, 1)[:5], 3)
    o = torch.zeros((2, 3, 10))
    o[:, :, -1] = torch.tensor([[0.1, 0.7, 0.7], [0.3, 0.2, 0.1]])
    sel, mask = P.filter_outputs(o)
    assert sel.shape[0] == 3 and mask.sum() == 3


# --------------------------------------------------------------------------- C ABI host-only mode
def _split(lib, x):
    hi = np.empty(x.size, np.uint16)
    lo = np.empty(x.size, np.uint16)
    u16 = ctypes.POINTER(ctypes.c_uint16)
    _lib.check(lib.ml_debug_split_f16(_lib.fptr(x), x.size, hi.ctypes.data_as(u16), lo.ctypes.data_as(u16)))
    return hi.view(np.float16), lo.view(np.float16)


def test_f16_split_matches_numpy(hip_lib):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(50000).astype(np.float32) * s for s in (1e-7, 1e-5, 1e-3, 1, 100, 3e4)]
                       + [np.array([0, -0.0, 65504, 65519.9, 65520, 7e4, -7e4, 1e-8, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5],
                                   dtype=np.float32)])
    hi, lo = _split(hip_lib, x)
    with np.errstate(over='ignore'):
        xc = np.clip(x, -65504, 65504)                    # the pair saturates: lo is taken from the CLAMPED value
        hi_ref = xc.astype(np.float16)
        lo_ref = (xc - hi_ref.astype(np.float32)).astype(np.float16)
    assert np.isfinite(lo.astype(np.float32)).all()
    assert np.array_equal(hi.view(np.uint16), hi_ref.view(np.uint16))
    assert np.array_equal(lo.view(np.uint16), lo_ref.view(np.uint16))
    ok = np.abs(x) < 6e4
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.all(np.abs(rec[ok] - x[ok]) <= np.maximum(np.abs(x[ok]) * 2.0 ** -21, 2.0 ** -25))


def _host_model(lib, sd, in_f, out_f, hidden, flags):
    h = ctypes.c_void_p()
    _lib.check(lib.ml_loco_create(in_f, hidden, out_f, 3, ctypes.byref(h)))
    for k, v in sd.items():
        a = np.ascontiguousarray(v, dtype=np.float32)
        _lib.check(lib.ml_loco_set_tensor(h, k.encode(), _lib.fptr(a), a.size))
    _lib.check(lib.ml_loco_finalize(h, _lib.ML_PREC_F16X2, flags | _lib.ML_FLAG_HOST_ONLY))
    return h


def _layer(lib, h, li):
    n, k, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ml_debug_get_layer(h, li, None, None, ctypes.byref(n), ctypes.byref(k), ctypes.byref(e)))
    w = np.empty((n.value, k.value), np.float32)
    b = np.empty(n.value, np.float32)
    _lib.check(lib.ml_debug_get_layer(h, li, _lib.fptr(w), _lib.fptr(b), None, None, None))
    kpad = (k.value + 31) // 32 * 32
    pk = np.empty(n.value * kpad * 2, np.uint16)
    _lib.check(lib.ml_debug_get_packed(h, li, pk.ctypes.data_as(ctypes.POINTER(ctypes.c_uint16)), pk.size))
    return w, b, e.value, pk.view(np.float16).reshape(n.value, kpad // 32, 2, 32)


def _head(lib, h, hi, hidden):
    nh, c0, sb, al = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.ml_debug_get_head(h, hi, None, None, ctypes.byref(nh), ctypes.byref(c0), ctypes.byref(sb),
                                     ctypes.byref(al)))
    w = np.empty((nh.value, hidden), np.float32)
    b = np.empty(nh.value, np.float32)
    _lib.check(lib.ml_debug_get_head(h, hi, _lib.fptr(w), _lib.fptr(b), None, None, None, None))
    return w, b, c0.value, sb.value, al.value


@pytest.mark.parametrize("merge,hidden", [(True, 256), (False, 256), (True, 200), (False, 300)])
def test_fold_merge_pack_emulated_forward(hip_lib, merge, hidden):
    """Fold + merge + pack in the library (host only), then emulate the kernels' arithmetic in numpy from
    the PACKED fp16 hi|lo images (3 products per term) and compare with the oracle: validates the
    whole host-side weight preparation and the precision design without a GPU."""
    from oracle import monoloco_oracle as O
    sd = synth.make_state_dict(5, 34, 9, hidden)
    h = _host_model(hip_lib, sd, 34, 9, hidden, _lib.ML_FLAG_MERGE_W2W3 if merge else 0)
    nl = hip_lib.ml_debug_num_layers(h)
    assert nl == (8 if merge else 9)
    layers = [_layer(hip_lib, h, i) for i in range(nl)]
    # any linear_size runs on 256-column tiles: zero weights/bias in the padded rows and columns
    hp = (hidden + 255) // 256 * 256
    assert layers[0][0].shape == (hp, 34) and layers[1][0].shape == (hp, hp)
    if hp != hidden:
        for w, b, _, _ in layers:
            assert not w[hidden:].any() and not b[hidden:].any() and (w.shape[1] == 34 or not w[:, hidden:].any())
    heads = [_head(hip_lib, h, i, hp) for i in range(2)]
    assert all(not hd[0][:, hidden:].any() for hd in heads)
    kps = torch.tensor(synth.make_poses(200, 3))
    ref = O.forward_mono({k: torch.tensor(v) for k, v in sd.items()}, kps, synth.KITTI_K)
    ref64 = O.forward_mono({k: torch.tensor(v) for k, v in sd.items()}, kps, synth.KITTI_K, dtype=torch.float64)

    def split(a):
        hi = a.astype(np.float16)
        return hi, (a - hi.astype(np.float32)).astype(np.float16)

    def dense(a, li, relu, res=None):
        w, b, e, pk = layers[li]
        k = w.shape[1]
        assert abs(np.abs(w).max() * 2.0 ** e) < 16384 * 1.0001 and np.abs(w).max() * 2.0 ** e >= 8192
        whi = pk[:, :, 0, :].reshape(w.shape[0], -1)[:, :k].astype(np.float64)
        wlo = pk[:, :, 1, :].reshape(w.shape[0], -1)[:, :k].astype(np.float64)
        assert np.abs((whi + wlo) * 2.0 ** -e - w).max() <= np.abs(w).max() * 2.0 ** -21
        ahi, alo = split(a)
        acc = (ahi.astype(np.float64) @ whi.T + ahi.astype(np.float64) @ wlo.T + alo.astype(np.float64) @ whi.T)
        v = acc.astype(np.float32) * np.float32(2.0 ** -e) + b
        if relu:
            v = np.maximum(v, 0)
        if res is not None:
            rh, rl = split(res)
            v = v + (rh.astype(np.float32) + rl.astype(np.float32))
        return v.astype(np.float32)

    bufs = {1: None, 2: None}
    a = dense(ref['inputs'].numpy(), 0, True)
    li = 1
    for _ in range(3):
        t = dense(a, li, True)
        a = dense(t, li + 1, True, res=a)
        li += 2
    bufs[1] = a
    raw = np.zeros((200, 9), np.float32)

    def run_head(hd, act):
        w, b, c0, _, _ = hd
        q = split(act)
        x = q[0].astype(np.float64) + q[1].astype(np.float64)
        raw[:, c0:c0 + w.shape[0]] = (x @ w.astype(np.float64).T + b).astype(np.float32)
    if merge:
        assert heads[0][3] == 1 and heads[0][4] == 6 and heads[1][3] == 2 and heads[1][4] == 7
        run_head(heads[0], a)
        y3 = dense(a, 7, True)
        run_head(heads[1], y3)
    else:
        assert heads[0][3] == 2 and heads[0][4] == 7 and heads[1][3] == 1 and heads[1][4] == 8
        y2 = dense(a, 7, False)
        run_head(heads[0], y2)
        y3 = dense(y2, 8, True)
        run_head(heads[1], y3)
    noise = np.abs(ref['raw'].numpy() - ref64['raw'].numpy()).max()
    assert np.abs(raw - ref64['raw'].numpy()).max() <= max(2 * noise, 1e-5)
    assert np.abs(raw - ref['raw'].numpy()).max() <= 5e-5
    hip_lib.ml_loco_destroy(h)


def test_epoch_batches_are_the_dataloaders():
    """Trainer's loader-free epoch sampler: the batches of DataLoader(n rows, batch_size, shuffle=True) -- the reference's loaders,
    trainer.py:104-106 -- and the same state of torch's global generator afterwards, epoch after epoch; its self-check is what
    decides at run time whether it is used."""
    from torch.utils.data import DataLoader
    from monoloco_amd.train.trainer import _EpochBatches, _IndexDataset
    for n, bs in ((331, 512), (1056, 64), (169, 512), (7, 3), (1, 1)):
        fast = _EpochBatches(n, bs)
        assert fast.matches_dataloader(epochs=3)
        torch.manual_seed(n)
        mine = [[b.tolist() for b in fast] for _ in range(2)]
        state_mine = torch.get_rng_state()
        torch.manual_seed(n)
        loader = DataLoader(_IndexDataset(n), batch_size=bs, shuffle=True)
        ref = [[b.tolist() for b in loader] for _ in range(2)]
        assert mine == ref and torch.equal(state_mine, torch.get_rng_state())
        assert sorted(i for b in mine[0] for i in b) == list(range(n))


def test_pyhost_list_walk_matches_numpy():
    """csrc/pyhost.c (optional host helper of Loco.forward): [m][3][17] nested lists straight into a float32 buffer -- the same
    values as np.asarray(dtype=float32), ints accepted, anything else refused (the caller then takes the numpy route)."""
    import ctypes
    import subprocess
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(['make', '-C', os.path.join(root, 'monoloco_amd', 'csrc'), 'pyhost'], check=True, stdout=subprocess.DEVNULL)
    from monoloco_amd.network import net as N
    N._PYHOST[0] = False
    ph = N._pyhost()
    assert ph is not None
    rng = np.random.default_rng(0)
    kp = (rng.random((16, 3, 17)) * 1000).tolist()
    kp[2][1][5] = 7                                            # a Python int among the floats
    dst = np.full((16, 3, 17), -1.0, dtype=np.float32)
    assert ph.ml_py_fill_kps(kp, dst.ctypes.data, 16) == 0
    assert np.array_equal(dst, np.asarray(kp, dtype=np.float32))
    assert ph.ml_py_fill_kps(kp, dst.ctypes.data, 15) == 1     # wrong person count
    bad = [list(p) for p in kp]
    bad[3] = bad[3][:2]                                        # ragged
    assert ph.ml_py_fill_kps(bad, dst.ctypes.data, 16) == 1
    bad = [list(p) for p in kp]
    bad[0] = [list(r) for r in bad[0]]
    bad[0][0][0] = 'x'
    assert ph.ml_py_fill_kps(bad, dst.ctypes.data, 16) == 1
    assert ph.ml_py_fill_kps(tuple(kp), dst.ctypes.data, 16) == 1   # not a list: numpy's job
    assert np.array_equal(N._lists_to_f32(kp), np.asarray(kp, dtype=np.float32))


def test_route_plan_per_row_count(hip_lib):
    """ml_loco_plan: the launch plan the forward executes (one make_plan in the library), asserted per row count -- which kernel family
    every dense layer takes and where the heads are fused."""
    sd = synth.make_state_dict(1, 34, 9, 1024)
    h = _host_model(hip_lib, sd, 34, 9, 1024, _lib.ML_FLAG_MERGE_W2W3)

    def plan(rows, mc=0, post=1):
        buf = ctypes.create_string_buffer(1024)
        _lib.check(hip_lib.ml_loco_plan(h, rows, mc, post, buf, 1024))
        return buf.value.decode()
    try:
        # the batch path: every head inside a dense epilogue, one tail launch
        assert plan(65536) == "route=tile; L0 pp; L1 w4; L2 w4; L3 w4; L4 w4; L5 w4; L6 w4+aux; L7 pp+fin8; end=tail_mono"
        assert plan(8193) == plan(65536) and plan(65536, post=0).endswith("end=reduce")
        # a single image: small tiles, both heads + post-process in the last launch
        assert plan(16) == "route=small16; " + "; ".join("L%d small16" % i for i in range(8)) + "; end=heads_small+post"
        assert plan(128).endswith("end=heads_small+post") and "small32" in plan(129) and plan(129).endswith("L6 small32 heads1; L7 small32 heads8; end=heads")
        # the mid window: dense_mid_kernel, from 4097 rows the half-size w4 tile for the long-K layers; round 5: both heads in the epilogues
        assert plan(2048) == "route=mid64; " + "; ".join("L%d mid64" % i for i in range(6)) + "; L6 mid64+aux; L7 mid64+fin8; end=tail_mono"
        assert plan(4096).startswith("route=mid128; L0 mid128; L1 mid128") and plan(4096).endswith("L6 mid128+aux; L7 mid128+fin8; end=tail_mono")
        assert plan(8192) == "route=half; L0 mid128; " + "; ".join("L%d half" % i for i in range(1, 6)) + "; L6 half+aux; L7 half+fin8; end=tail_mono"
        assert plan(8192, post=0).endswith("end=reduce")
        _lib.check(hip_lib.ml_loco_set_option(h, b"mid_heads", 0))          # rounds 3-4: the pair kernel behind the last layer
        assert plan(2048).endswith("L6 mid64; L7 mid64; end=heads_pair+post") and plan(8192).endswith("L7 half; end=heads_pair+post")
        _lib.check(hip_lib.ml_loco_set_option(h, b"mid_heads", 1))
        assert hip_lib.ml_loco_set_option(h, b"no_such_option", 1) == 1
        # a stochastic pass: no head fusion, masks behind layer 0 and in front of w_fin (reference net.py:141)
        assert plan(65536, mc=1, post=0) == "route=tile; L0 pp+dropout; L1 w4; L2 w4; L3 w4; L4 w4; L5 w4; L6 w4 heads1; L7 w4+dropout heads8; end=heads"
        assert hip_lib.ml_loco_route(h, 8192) == 4 and hip_lib.ml_loco_route(h, 8193) == 5
        # tuning moves the plan, nothing else does
        _lib.check(hip_lib.ml_loco_set_tuning(h, -1, -1, -1, 2, -1, -1))     # dense_kernel_pp everywhere: the aux head cannot ride
        assert plan(65536) == "route=tile; " + "; ".join("L%d pp" % i for i in range(6)) + "; L6 pp heads1; L7 pp+fin8; end=reduce"
        assert hip_lib.ml_loco_plan(h, 16, 0, 1, ctypes.create_string_buffer(8), 8) == 1   # ML_ERR_ARG: buffer too small
    finally:
        hip_lib.ml_loco_destroy(h)


def test_preprocess_mask_reads_the_mask_annotations(tmp_path):
    """reference process.py:136-152: `<parent>/mask[_right]/<basename>.json` -> (boxes, [[xs, ys, cs]]); a missing file gives ([], [])."""
    import json
    from monoloco_amd.network.process import preprocess_mask
    kps = [[float(3 * j + c) for j in range(17) for c in range(3)]]
    for sub in ('mask', 'mask_right'):
        (tmp_path / sub).mkdir()
        (tmp_path / sub / 'img.json').write_text(json.dumps({'boxes': [[1, 2, 3, 4]], 'keypoints': kps}))
    for mode in ('left', 'right'):
        boxes, keypoints = preprocess_mask(str(tmp_path / 'annotations'), 'img', mode)
        assert boxes == [[1, 2, 3, 4]] and keypoints == [[kps[0][0::3], kps[0][1::3], kps[0][2::3]]]
    assert preprocess_mask(str(tmp_path / 'annotations'), 'nothing') == ([], [])


def test_modules_copy_and_pickle_without_their_native_handles():
    """copy.deepcopy(model) and torch.save(model) work whatever native caches the module holds (the packed eval engine, the train-mode
    twin): they are raw-pointer handles, dropped from the copied state and rebuilt on first use."""
    import copy
    import ctypes
    import io
    import torch
    from monoloco_amd.network.architectures import LocoModel

    class Handle:
        def __init__(self):
            self._h = ctypes.c_void_p(5)
    m = LocoModel(34, 9, 256)
    m._hip_tr, m._engine = Handle(), Handle()
    m2 = copy.deepcopy(m)
    assert m2._engine is None and m2._hip_tr is None and torch.equal(m2.w1.weight, m.w1.weight)
    torch.save(m, io.BytesIO())
