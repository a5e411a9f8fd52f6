"""Ground-truth association (the batched half of post_process) on the CPU: the oracle's restatement and the
product's native host routines (csrc/matching.hip, host entry points -- no GPU involved) against results of the REAL
reference on seeded box sets with 16 / 256 / 2048 detections (tests/golden/golden_matching.json, oracle/make_golden.py
`matching`), ties included.  The device kernels of the same file are checked in tests/test_gpu_matching.py."""
import json
import os
import time

import numpy as np
import pytest

import synth
from monoloco_amd.utils import iou as I
from oracle import monoloco_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
GOLD = json.load(open(os.path.join(G, 'golden_matching.json')))
CASES = GOLD['cases']
IDS = ['%dx%d%s' % (c['m'], c['g'], '_ties' if c['ties'] else '') for c in CASES]


def same_sort_as_golden(case, boxes):
    """True when this machine's np.argsort gives the order the golden run saw (an unstable sort's tie order may depend on
    the CPU's SIMD dispatch; without ties it always does)."""
    return (np.argsort([b[4] for b in boxes]).tolist() == case['argsort_conf']
            and np.argsort([b[0] for b in boxes]).tolist() == case['argsort_left'])


def pairs(lst):
    return [tuple(p) for p in lst]


@pytest.fixture
def host_only(monkeypatch):
    """Every size through the host entry points (the CPU suite has no device; the arithmetic is the same source)."""
    monkeypatch.setattr(I, 'DEVICE_MIN_PAIRS', 1 << 62)


@pytest.mark.parametrize("case", [c for c in CASES if c['m'] <= 256], ids=[i for c, i in zip(CASES, IDS) if c['m'] <= 256])
def test_oracle_matching_is_the_references(case):
    boxes, gt = synth.make_boxes(case['m'], case['g'], case['seed'], ties=case['ties'])
    if not same_sort_as_golden(case, boxes):
        pytest.skip("np.argsort breaks ties differently on this CPU than where the golden was made")
    matches = O.get_iou_matches(boxes, gt, case['iou_min'])
    assert matches == pairs(case['matches'])
    assert O.reorder_matches(matches, boxes) == pairs(case['ordered'])
    mat = O.get_iou_matrix(boxes, gt)
    assert float(mat.sum()) == case['iou_sum'] and mat[3].tolist() == case['iou_row3']


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_host_matching_is_the_references(hip_lib, host_only, case):
    boxes, gt = synth.make_boxes(case['m'], case['g'], case['seed'], ties=case['ties'])
    matches = I.get_iou_matches(boxes, gt, case['iou_min'])
    ordered = I.get_iou_matches_ordered(boxes, gt, case['iou_min'])
    assert all(type(v) is int for p in matches for v in p) and all(type(p) is tuple for p in matches)
    assert I.reorder_matches(matches, boxes, mode='left_right') == ordered
    if same_sort_as_golden(case, boxes):
        assert matches == pairs(case['matches'])
        assert ordered == pairs(case['ordered'])
    elif case['m'] <= 256:   # another tie order on this CPU: the oracle makes the same np.argsort calls here
        assert matches == O.get_iou_matches(boxes, gt, case['iou_min'])
        assert ordered == O.reorder_matches(matches, boxes)
    if 'iou_sum' in case:
        mat = I.get_iou_matrix(boxes, gt)
        assert mat.dtype == np.float64 and mat.shape == (case['m'], case['g'])
        assert float(mat.sum()) == case['iou_sum'] and mat[3].tolist() == case['iou_row3']
        assert [(int(a), int(b)) for a, b in I.get_iou_matches_matrix(boxes, gt, case['iou_min'])] == pairs(case['matches_matrix'])


def test_matching_edge_cases(hip_lib):
    boxes, gt = synth.make_boxes(12, 9, 3)
    assert I.get_iou_matches([], gt) == [] and I.get_iou_matches(boxes, []) == [] and I.get_iou_matches_ordered([], []) == []
    assert I.get_iou_matrix([], gt).shape == (0, 9) and I.get_iou_matrix(boxes, []).shape == (12, 0)
    # ground-truth rows of unequal length (a trailing field on some): only x1, y1, x2, y2 are read
    ragged = [row + [1.0] if i % 2 else row for i, row in enumerate(gt)]
    assert I.get_iou_matches(boxes, ragged) == I.get_iou_matches(boxes, gt) == O.get_iou_matches(boxes, gt)
    # a zero union is the reference's ZeroDivisionError (iou.py:25), not a silent nan
    with pytest.raises(ZeroDivisionError):
        I.get_iou_matches([[5., 5., 5., 5., 0.9]], [[5., 5., 5., 5.]])
    with pytest.raises(ZeroDivisionError):
        O.get_iou_matches([[5., 5., 5., 5., 0.9]], [[5., 5., 5., 5.]])
    with pytest.raises(ZeroDivisionError):
        I.get_iou_matrix([[5., 5., 5., 5., 0.9]], [[5., 5., 5., 5.]])
    # ragged DETECTIONS (some carry fields behind the confidence): box[4] still orders the greedy pass, as in the reference's loop
    ragged_det = [row + [7.0, 8.0] if i % 3 == 0 else row for i, row in enumerate(boxes)]
    assert I.get_iou_matches(ragged_det, gt) == O.get_iou_matches(boxes, gt)
    assert I.get_iou_matches_ordered(ragged_det, ragged) == O.reorder_matches(O.get_iou_matches(boxes, gt), boxes)
    # numpy scalars instead of Python floats (make_lower_boxes-style arrays): the reference's division gives nan + a RuntimeWarning
    # there, no exception -- same here; the nan never passes `>= iou_min`
    with np.errstate(invalid='ignore', divide='ignore'):
        want = O.get_iou_matrix(np.array([[5., 5., 5., 5., 0.9]]), np.array([[5., 5., 5., 5.]]))
    got = I.get_iou_matrix(np.array([[5., 5., 5., 5., 0.9]]), np.array([[5., 5., 5., 5.]]))
    assert np.isnan(want).all() and np.isnan(got).all()
    np_rows = lambda rows: [[np.float64(v) for v in row] for row in rows]     # (an ndarray fails `not boxes` in the reference too)
    with np.errstate(invalid='ignore', divide='ignore'):
        assert I.get_iou_matches(np_rows([[5., 5., 5., 5., 0.9]]), np_rows([[5., 5., 5., 5.]])) == [] \
            == O.get_iou_matches(np_rows([[5., 5., 5., 5., 0.9]]), np_rows([[5., 5., 5., 5.]]))
    # a box set large enough for the device kernels on a box WITHOUT a device: the library's host loops (a CPU-only dataset-preparation
    # or evaluation box that re-binds these helpers through compat.install())
    import torch
    if not torch.cuda.is_available():
        bb, gg = synth.make_boxes(256, 256, 11)
        assert 256 * 256 >= I.DEVICE_MIN_PAIRS
        assert I.get_iou_matches(bb, gg) == O.get_iou_matches(bb, gg)
        assert np.array_equal(I.get_iou_matrix(bb[:200], gg[:200] + gg[:40]), O.get_iou_matrix(bb[:200], gg[:200] + gg[:40]))
    # a NaN IoU is np.argmax's first choice and never passes `>= iou_min`
    nan_gt = [[float('nan'), 0., 10., 10.], [0., 0., 10., 10.]]
    assert I.get_iou_matches([[0., 0., 10., 10., 0.5]], nan_gt) == O.get_iou_matches([[0., 0., 10., 10., 0.5]], nan_gt)
    # threshold is inclusive; iou exactly 1.0
    assert I.get_iou_matches([[0., 0., 10., 10., 0.5]], [[0., 0., 10., 10.]], iou_min=1.0) == [(0, 0)]
    # reorder_matches on lists the greedy pass cannot produce: an index twice (first match wins), indices outside the boxes
    weird = [(3, 1), (0, 2), (3, 7), (40, 0), (-1, 5), (2, 2)]
    assert I.reorder_matches(weird, boxes, mode='left_right') == O.reorder_matches(weird, boxes)
    assert I.reorder_matches([], boxes, mode='left_right') == []
    with pytest.raises(AssertionError):
        I.reorder_matches(weird, boxes, mode='left_rigth')   # the reference's default value is a typo its own assert rejects


def test_host_matching_cost(hip_lib, host_only):
    """The quadratic Python of the reference took 0.5 ms at 16 boxes and 5.4 s at 2048: bounds far above what
    the native pass needs, far below that."""
    for m, bound in ((16, 2e-3), (2048, 0.5)):
        boxes, gt = synth.make_boxes(m, m, 5)
        I.get_iou_matches_ordered(boxes, gt)
        times = []
        for _ in range(3):   # (the best of three: a loaded host must not fail a cost bound that is 10 x off either way)
            t0 = time.perf_counter()
            I.get_iou_matches_ordered(boxes, gt)
            times.append(time.perf_counter() - t0)
        assert min(times) < bound, times


def test_matching_property_random_boxes_with_many_ties(hip_lib, host_only):
    """Hypothesis: boxes on a coarse integer grid (IoU ties and 0 / 1 values galore), confidences from a 4-value set (argsort ties),
    any iou_min -- the native pass equals the oracle's loops (same np.argsort calls, so the same tie order on this machine)."""
    from hypothesis import given, settings, strategies as st

    coord = st.integers(0, 12)

    @st.composite
    def box(draw, with_conf):
        x1, y1 = draw(coord), draw(coord)
        w, h = draw(st.integers(1, 6)), draw(st.integers(1, 6))
        b = [float(x1), float(y1), float(x1 + w), float(y1 + h)]
        return b + [draw(st.sampled_from([0.2, 0.5, 0.5, 0.9]))] if with_conf else b

    @settings(max_examples=300, deadline=None)
    @given(st.lists(box(True), min_size=1, max_size=24), st.lists(box(False), min_size=1, max_size=24),
           st.sampled_from([0.0, 0.25, 0.3, 0.5, 1.0]))
    def check(boxes, gt, iou_min):
        want = O.get_iou_matches(boxes, gt, iou_min)
        assert I.get_iou_matches(boxes, gt, iou_min) == want
        assert I.get_iou_matches_ordered(boxes, gt, iou_min) == O.reorder_matches(want, boxes)
        assert np.array_equal(I.get_iou_matrix(boxes, gt), O.get_iou_matrix(boxes, gt))
    check()
