"""Box matching helpers used by ``Loco.post_process`` when ground truth is supplied (reference
monoloco/utils/iou.py).  Pure host logic on a handful of boxes; re-written, not part of the
device path."""
import json

import numpy as np


def calculate_iou(box1, box2):
    """IoU of two (x1, y1, x2, y2[, ...]) boxes: inter / (area1 + area2 - inter), disjoint boxes give 0
    (reference iou.py:6-28)."""
    iw = max(min(box1[2], box2[2]) - max(box1[0], box2[0]), 0)
    ih = max(min(box1[3], box2[3]) - max(box1[1], box2[1]), 0)
    inter = iw * ih
    area1 = (box1[2] - box1[0]) * (box1[3] - box1[1])
    area2 = (box2[2] - box2[0]) * (box2[3] - box2[1])
    return inter / (area1 + area2 - inter)


def get_iou_matrix(boxes, boxes_gt):
    """(len(boxes), len(boxes_gt)) IoU matrix (reference iou.py:31-41)."""
    mat = np.zeros((len(boxes), len(boxes_gt)))
    for i, box in enumerate(boxes):
        for j, gt in enumerate(boxes_gt):
            mat[i, j] = calculate_iou(box, gt)
    return mat


def get_iou_matches(boxes, boxes_gt, iou_min=0.3):
    """Visit detections by decreasing confidence (box[4]); each looks at its best-IoU ground-truth box
    (over ALL gt boxes) and is matched only if that IoU >= iou_min and the gt box is still free
    (reference iou.py:44-64).  Returns [(idx, idx_gt), ...] in visiting order."""
    if not boxes or not boxes_gt:
        return []
    matches, taken = [], set()
    for idx in reversed(list(np.argsort([b[4] for b in boxes]))):
        ious = [calculate_iou(boxes[idx], gt) for gt in boxes_gt]
        j = int(np.argmax(ious))
        if ious[j] >= iou_min and j not in taken:
            matches.append((int(idx), j))
            taken.add(j)
    return matches


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
    (reference iou.py:86-100)."""
    assert mode == 'left_right'
    matched = [int(i) for i, _ in matches]
    return [matches[matched.index(i)] for i in np.argsort([b[0] for b in boxes]) if i in matched]


def open_annotations(path_ann):
    try:
        with open(path_ann, 'r') as f:
            return json.load(f)
    except FileNotFoundError:
        return []
