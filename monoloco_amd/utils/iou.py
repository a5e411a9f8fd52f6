"""Box matching helpers used by ``Loco.post_process`` when ground truth is supplied (reference
monoloco/utils/iou.py), for a whole image at a time.

The reference calls a scalar ``calculate_iou`` m x g times from Python and searches lists inside loops (quadratic to
cubic in the number of boxes; ``GenerateKitti`` passes ground truth on every image).  Here the IoUs of all pairs are
computed in ONE native call -- ``csrc/matching.hip``: plain host loops for an image with a handful of boxes, a gfx950
kernel (one wavefront per detection, fp64) from ``DEVICE_MIN_PAIRS`` pairs on -- and the greedy / re-ordering passes are
linear.  Every value is an IEEE double computed in the order of the reference's Python expressions, with Python's
``max`` / ``min`` and ``np.argmax`` tie rules and the very same ``np.argsort`` calls, so the matches are the reference's
own, ties included.  There is no fallback: a missing library raises (``_lib.load``)."""
import ctypes
import json

import numpy as np

from .. import _lib

# from this many (detection, ground-truth) pairs on, the IoUs are computed on the GPU (below it a launch, two uploads
# and a read-back cost more than the host loop: 256 IoUs of a 16 x 16 frame take ~1 us)
DEVICE_MIN_PAIRS = 1 << 15


def _check(code):
    if code != 0:
        msg = _lib.load().ml_matching_last_error()
        raise _lib.MonolocoHipError("monoloco_hip matching error %d: %s" % (code, msg.decode() if msg else '?'))


def _boxes_f64(boxes, cols=4):
    """Rows of Python numbers -> one C-contiguous (n, >=cols) float64 array (a Python float IS that double).  Rows of
    unequal length (ground truth with and without a trailing field, detections that carry extra fields) keep their first
    `cols` columns: 4 for ground truth, 5 for detections whose confidence (box[4]) orders the greedy pass."""
    try:
        arr = np.asarray(boxes, dtype=np.float64)
    except ValueError:
        arr = None
    if arr is None or arr.ndim != 2:
        arr = np.asarray([box[:cols] for box in boxes], dtype=np.float64)
    assert arr.ndim == 2 and arr.shape[1] >= cols, "boxes must be rows of x1, y1, x2, y2[, conf, ...]"
    return arr if arr.flags.c_contiguous else np.ascontiguousarray(arr)


def _python_floats(*box_sets):
    """True when every box set holds plain Python numbers.  The reference's calculate_iou divides whatever it is handed
    (iou.py:25): Python floats raise ZeroDivisionError on a zero union, numpy scalars (e.g. make_lower_boxes' arrays) give
    nan / inf and a RuntimeWarning instead -- the native calls compute the IEEE quotient either way and report the event."""
    for boxes in box_sets:
        if isinstance(boxes, np.ndarray):
            return False
        if len(boxes) and (isinstance(boxes[0], np.ndarray) or isinstance(boxes[0][0], np.generic)):
            return False
    return True


def _use_device(pairs):
    """The IoUs of an image run on the GPU from DEVICE_MIN_PAIRS pairs on -- when there is one: a dataset-preparation or
    evaluation box without a HIP device (compat.install() re-binds these helpers into the reference's eval / prep modules)
    takes the library's host loops at every size (the same doubles in the same order, tests/test_matching.py)."""
    if pairs < DEVICE_MIN_PAIRS:
        return False
    import torch
    return torch.cuda.is_available()


def _raise_zero_div(flag, python_floats=True):
    if flag and python_floats:
        raise ZeroDivisionError("float division by zero")   # what the reference's calculate_iou raises (iou.py:25)


def calculate_iou(box1, box2):
    """IoU of two (x1, y1, x2, y2[, ...]) boxes: inter / (area1 + area2 - inter), disjoint boxes give 0
    (reference iou.py:6-28).  One pair; the set-wise functions below do not go through here."""
    iw = max(min(box1[2], box2[2]) - max(box1[0], box2[0]), 0)
    ih = max(min(box1[3], box2[3]) - max(box1[1], box2[1]), 0)
    inter = iw * ih
    area1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    area2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    return inter / (area1 + area2 - inter)


def _device_buffers(b, gt, out_bytes):
    """Uploads of the two box arrays + one zeroed byte buffer for the results, on the current HIP device."""
    import torch
    from .. import engine
    dev = engine._require_cuda(None)
    b_d = torch.from_numpy(b).to(dev)
    gt_d = torch.from_numpy(gt).to(dev)
    out_d = torch.zeros((out_bytes,), dtype=torch.uint8, device=dev)
    return dev, b_d, gt_d, out_d


def _best_rows(b, gt, python_floats=True):
    """(jmax int32 (m), vmax float64 (m)): per detection the first arg-max of the IoU over all ground-truth boxes."""
    lib = _lib.load()
    m, g = b.shape[0], gt.shape[0]
    if not _use_device(m * g):
        jmax = np.empty((m,), dtype=np.int32)
        vmax = np.empty((m,), dtype=np.float64)
        flag = ctypes.c_int32(0)
        _check(lib.ml_iou_best_host(b.ctypes.data, m, b.shape[1], gt.ctypes.data, g, gt.shape[1], jmax.ctypes.data,
                                    vmax.ctypes.data, ctypes.byref(flag)))
        _raise_zero_div(flag.value, python_floats)
        return jmax, vmax
    import torch
    from .. import engine
    # one result buffer, one copy back: vmax (m doubles) | jmax (m int32) | zero-division flag (int32)
    dev, b_d, gt_d, out_d = _device_buffers(b, gt, 12 * m + 4)
    base = out_d.data_ptr()
    with torch.cuda.device(dev):
        _check(lib.ml_iou_best(b_d.data_ptr(), m, b.shape[1], gt_d.data_ptr(), g, gt.shape[1], base + 8 * m, base,
                               base + 12 * m, engine._stream(dev)))
    host = out_d.cpu().numpy()
    vmax = host[:8 * m].view(np.float64)
    jmax = host[8 * m:12 * m].view(np.int32)
    _raise_zero_div(int(host[12 * m:].view(np.int32)[0]), python_floats)
    return jmax, vmax


def get_iou_matrix(boxes, boxes_gt):
    """(len(boxes), len(boxes_gt)) IoU matrix (reference iou.py:31-41)."""
    m, g = len(boxes), len(boxes_gt)
    if m == 0 or g == 0:
        return np.zeros((m, g))
    lib = _lib.load()
    b, gt = _boxes_f64(boxes), _boxes_f64(boxes_gt)
    pyf = _python_floats(boxes, boxes_gt)
    if not _use_device(m * g):
        mat = np.empty((m, g), dtype=np.float64)
        flag = ctypes.c_int32(0)
        _check(lib.ml_iou_matrix_host(b.ctypes.data, m, b.shape[1], gt.ctypes.data, g, gt.shape[1], mat.ctypes.data,
                                      ctypes.byref(flag)))
        _raise_zero_div(flag.value, pyf)
        return mat
    import torch
    from .. import engine
    mat = np.empty((m, g), dtype=np.float64)
    dev, b_d, gt_d, flag_d = _device_buffers(b, gt, 4)
    rows = max(1, min(m, 65535, (1 << 28) // (8 * g)))   # detection rows per launch: <= 256 MB of IoUs in flight
    out_d = torch.empty((rows, g), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        for lo in range(0, m, rows):
            n = min(rows, m - lo)
            _check(lib.ml_iou_matrix(b_d.data_ptr() + lo * b.shape[1] * 8, n, b.shape[1], gt_d.data_ptr(), g, gt.shape[1],
                                     out_d.data_ptr(), flag_d.data_ptr(), engine._stream(dev)))
            mat[lo:lo + n] = out_d[:n].cpu().numpy()
    _raise_zero_div(int(flag_d.cpu().numpy().view(np.int32)[0]), pyf)
    return mat


def _matches(boxes, boxes_gt, iou_min, left_to_right):
    """get_iou_matches, optionally followed by reorder_matches, on arrays: one native call for the pairs."""
    lib = _lib.load()
    b, gt = _boxes_f64(boxes, cols=5), _boxes_f64(boxes_gt)
    pyf = _python_floats(boxes, boxes_gt)
    m, g = b.shape[0], gt.shape[0]
    # the reference's own sort calls on the same doubles (iou.py:51-53, :97): ties come out in numpy's order
    # (argsort of a strided column sorts a contiguous copy of it: the same call on the same values)
    order = np.argsort(b[:, 4])[::-1].copy()
    left = np.argsort(b[:, 0]) if left_to_right else None
    p_left = left.ctypes.data if left_to_right else None
    out = np.empty((2 * min(m, g) + 2,), dtype=np.int64)   # pairs | n_pairs | zero-division flag
    p_out = out.ctypes.data
    p_n, p_flag = p_out + 16 * min(m, g), p_out + 16 * min(m, g) + 8
    out[-1] = 0
    if not _use_device(m * g):
        _check(lib.ml_iou_matches_host(b.ctypes.data, m, b.shape[1], gt.ctypes.data, g, gt.shape[1], order.ctypes.data,
                                       float(iou_min), p_left, p_out, p_n, p_flag))
        _raise_zero_div(out[-1], pyf)
    else:
        jmax, vmax = _best_rows(b, gt, pyf)
        jmax, vmax = np.ascontiguousarray(jmax), np.ascontiguousarray(vmax)
        _check(lib.ml_iou_greedy(order.ctypes.data, m, jmax.ctypes.data, vmax.ctypes.data, m, g, float(iou_min), p_left,
                                 p_out, p_n))
    return list(map(tuple, out[:2 * int(out[-2])].reshape(-1, 2).tolist()))


def get_iou_matches(boxes, boxes_gt, iou_min=0.3):
    """Visit detections by decreasing confidence (box[4]); each looks at its best-IoU ground-truth box
    (over ALL gt boxes) and is matched only if that IoU >= iou_min and the gt box is still free
    (reference iou.py:44-64).  Returns [(idx, idx_gt), ...] in visiting order."""
    if not boxes or not boxes_gt:
        return []
    return _matches(boxes, boxes_gt, iou_min, False)


def get_iou_matches_ordered(boxes, boxes_gt, iou_min=0.3):
    """reorder_matches(get_iou_matches(boxes, boxes_gt, iou_min), boxes, mode='left_right') -- what Loco.post_process
    needs (reference net.py:173, 187-188) -- without building the intermediate list."""
    if not boxes or not boxes_gt:
        return []
    return _matches(boxes, boxes_gt, iou_min, True)


def get_iou_matches_matrix(boxes, boxes_gt, thresh):
    """Repeatedly take the global maximum of the IoU matrix while it exceeds thresh, then retire its
    row and column (reference iou.py:67-83)."""
    mat = get_iou_matrix(boxes, boxes_gt)
    if not mat.size:
        return []
    matches = []
    while mat.max() > thresh:
        i, j = np.unravel_index(np.argmax(mat, axis=None), mat.shape)
        matches.append((i, j))
        mat[i, :] = 0
        mat[:, j] = 0
    return matches


def reorder_matches(matches, boxes, mode='left_right'):
    """Matched detections re-ordered by the left edge (box[0]) of their box, left to right
    (reference iou.py:86-100): for every box in np.argsort order that occurs among the matches, the FIRST match
    holding it -- one scatter and one gather instead of a list search per box."""
    assert mode == 'left_right'
    n = len(boxes)
    ordered = np.argsort([box[0] for box in boxes])   # the reference's own call (iou.py:97)
    if not matches or n == 0:
        return []
    left = np.asarray([int(idx) for idx, _ in matches], dtype=np.int64)
    where = np.full((n,), -1, dtype=np.int64)
    inside = np.flatnonzero((left >= 0) & (left < n))[::-1]   # last to first: the first match of a box is written last
    where[left[inside]] = inside
    sel = where[ordered]
    return [matches[k] for k in sel[sel >= 0].tolist()]


def open_annotations(path_ann):
    try:
        with open(path_ann, 'r') as f:
            return json.load(f)
    except FileNotFoundError:
        return []
