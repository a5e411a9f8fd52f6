"""Host mirror of ``monoloco.network.process`` for the keypoint->3D path.

Same names, argument meaning, return types and side effects as the reference
(monoloco/network/process.py); the JSON/list helpers are plain host Python (as in the
reference), every tensor computation runs in the HIP kernels behind the C ABI.
"""
import json
import logging
import os

import numpy as np
import torch
import yaml

from .. import engine
from .._lib import OUT_COLS as _C

logger = logging.getLogger(__name__)

SENSOR_W_MM = 7.2  # nuScenes sensor size used by the 'custom' calibration (reference process.py:20-21)
SENSOR_H_MM = 5.4


# ----------------------------------------------------------------------------- host-side JSON helpers
def prepare_pif_kps(kps_in):
    """Flat [x0, y0, c0, x1, ...] (51) -> [[x...], [y...], [c...]] (reference process.py:210-218)."""
    assert len(kps_in) % 3 == 0, "keypoints expected as a multiple of 3"
    return [kps_in[0::3], kps_in[1::3], kps_in[2::3]]


def preprocess_pifpaf(annotations, im_size=None, enlarge_boxes=True, min_conf=0.):
    """OpenPifPaf annotations -> (boxes [x1,y1,x2,y2,conf], keypoints [3][17]) (reference
    process.py:155-207).  Like the reference it edits each annotation's 'bbox' list IN PLACE and
    appends the confidence to it.  With a 'score' key the bbox is (x, y, w, h) and grows by h/10,
    w/5; without, it is already corners, conf = mean keypoint confidence and it grows by 1/7 of the
    height and 1/3.5 of the width; `enlarge_boxes=False` halves the growth."""
    boxes, keypoints = [], []
    shrink = 1 if enlarge_boxes else 2
    for ann in annotations:
        kps = prepare_pif_kps(ann['keypoints'])
        box = ann['bbox']
        if 'score' in ann:
            conf = ann['score']
            dh = box[3] / (10 * shrink)
            dw = box[2] / (5 * shrink)
            box[2] += box[0]
            box[3] += box[1]
        else:
            conf = float(np.mean(np.array(kps[2])))
            dh = (box[3] - box[1]) / (7 * shrink)
            dw = (box[2] - box[0]) / (3.5 * shrink)
            assert dh > -5 and dw > -5, "Bounding box <=0"
        box[0] -= dw
        box[1] -= dh
        box[2] += dw
        box[3] += dh
        if im_size is not None:
            box[0] = max(0, box[0])
            box[1] = max(0, box[1])
            box[2] = min(box[2], im_size[0])
            box[3] = min(box[3], im_size[1])
        if conf >= min_conf:
            box.append(conf)
            boxes.append(box)
            keypoints.append(kps)
    return boxes, keypoints


def preprocess_mask(dir_ann, basename, mode='left'):
    """Boxes and keypoints of the instance-mask annotations next to a PifPaf annotation directory (reference process.py:136-152;
    a host-side file reader off the keypoint -> 3D path, kept so that callers of the reference's `process` module find it):
    `<parent of dir_ann>/mask[_right]/<basename>.json` -> (boxes, [[xs, ys, cs], ...]); ([], []) when the file is missing."""
    from ..utils.iou import open_annotations
    mask_dir = os.path.join(os.path.split(dir_ann)[0], 'mask')
    assert mode in ('left', 'right'), "mode not recognized"
    path_ann = os.path.join(mask_dir if mode == 'left' else mask_dir + '_right', basename + '.json')
    dic = open_annotations(path_ann)
    if isinstance(dic, list):
        return [], []
    keypoints = [prepare_pif_kps(np.array(kps).reshape(51,).tolist()) for kps in dic['keypoints']]
    return dic['boxes'], keypoints


def image_transform(image):
    """ImageNet normalisation of a PIL image for the OpenPifPaf CNN (reference process.py:221-228): the CNN is out of scope here, the
    helper only forwards to torchvision when the caller's environment has it."""
    try:
        import torchvision
    except ImportError as exc:
        raise ImportError("image_transform feeds the OpenPifPaf CNN (outside the keypoint -> 3D path) and needs torchvision") from exc
    normalize = torchvision.transforms.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
    return torchvision.transforms.Compose([torchvision.transforms.ToTensor(), normalize])(image)


def load_calibration(calibration, im_size, focal_length=5.7):
    """Intrinsic matrix as nested lists (reference process.py:70-86): 'custom' derives it from the
    image size and a focal length in mm, otherwise intrinsics.yaml rescaled to im_size."""
    if calibration == 'custom':
        kk = [[im_size[0] * focal_length / SENSOR_W_MM, 0., im_size[0] / 2],
              [0., im_size[1] * focal_length / SENSOR_H_MM, im_size[1] / 2],
              [0., 0., 1.]]
    else:
        with open(os.path.join(os.path.dirname(os.path.realpath(__file__)), 'intrinsics.yaml')) as f:
            cfg = yaml.safe_load(f)[calibration]
        kk = cfg['intrinsics']
        sx, sy = (size / orig for size, orig in zip(im_size, cfg['im_size']))
        kk[0] = [el * sx for el in kk[0]]
        kk[1] = [el * sy for el in kk[1]]
    logger.info("Using %s calibration matrix", calibration)
    return kk


def factory_for_gt(path_gt, name=None):
    """(ground-truth dict of image `name`, its K) from a names-*.json file (reference process.py:89-98)."""
    assert os.path.exists(path_gt), "Ground-truth file not found"
    with open(path_gt, 'r') as f:
        dic_names = json.load(f)
    return dic_names[name], dic_names[name]['K']


# ----------------------------------------------------------------------------- tensor functions (HIP)
def _home(x):
    return x.device if isinstance(x, torch.Tensor) else torch.device('cpu')


def preprocess_monoloco(keypoints, kk, zero_center=False):
    """(m,3,17) pixel keypoints -> (m,34) normalised inputs at z = 10 m (reference process.py:47-67)."""
    home = _home(keypoints)
    x = engine.preprocess_mono(keypoints, kk.tolist() if isinstance(kk, torch.Tensor) else kk,
                               device=home if home.type == 'cuda' else None, zero_center=zero_center)
    return x.to(home)


def preprocess_monstereo(keypoints, keypoints_r, kk):
    """All-vs-all left x right rows [L_i, L_i - R_j] (reference process.py:25-44) -> (inputs, clusters)."""
    home = _home(keypoints)
    kk = kk.tolist() if isinstance(kk, torch.Tensor) else kk
    dev = home if home.type == 'cuda' else None
    xl = engine.preprocess_mono(keypoints, kk, device=dev)
    xr = engine.preprocess_mono(keypoints_r, kk, device=xl.device)
    rows = engine.stereo_pairs(xl, xr)
    return rows.to(home), [xr.shape[0]] * xl.shape[0]


def preprocess_monoloco_rows(keypoints, kks, k_index, keypoints_r=None):
    """Batched dataset preparation (reference prep/preprocess_kitti.py:190-253 runs preprocess_monoloco once per
    matched annotation): row i is normalised with kks[k_index[i]]; with `keypoints_r` the rows are the stereo
    training inputs [L, L - R].  One kernel launch, bit-identical to the per-annotation calls."""
    home = _home(keypoints)
    x = engine.preprocess_rows(keypoints, kks, k_index, kps_r=keypoints_r, device=home if home.type == 'cuda' else None)
    return x.to(home)


def unnormalize_bi(loc):
    """bi = exp(loc[:,1]) * loc[:,0] for loc (m,2) = (d, log b/d) (reference process.py:125-133)."""
    assert loc.size()[1] == 2, "size of the output tensor should be (m, 2)"
    home = loc.device
    raw = torch.zeros((loc.shape[0], 9), dtype=torch.float32, device=loc.device)
    raw[:, 2:4] = loc
    out, _ = engine.extract_outputs_device(raw.to(engine._require_cuda(home if home.type == 'cuda' else None)))
    return out[:, _C['bi']:_C['bi'] + 1].to(home)


def packed_to_dict(out, n_cols, home=torch.device('cpu')):
    """The (m,16) packed device result -> the reference's extract_outputs dictionary of (m,k) tensors,
    in the reference's key order (process.py:240-278).  One copy off the device, then one contiguous re-pack
    whose column groups become the (independent-looking, non-overlapping) output tensors."""
    o = out.to(home)
    # columns regrouped so that every output is a contiguous slice of its own row block
    cols = [8, 9, 10, 12, 13, 7, 4, 0, 1, 2, 3, 5, 6]   # h w l | ori0 ori1 | aux | bi | x y z d | yaw | yaw_ego
    t = o[:, cols].t().contiguous()                       # (13, m): each row one output column
    col = lambda a, b: t[a:b].t()                         # (m, b-a) view of rows a..b (non-overlapping storage)
    dic = {'h': col(0, 1), 'w': col(1, 2), 'l': col(2, 3), 'ori': col(3, 5).contiguous()}
    if n_cols == 10:
        dic['aux'] = col(5, 6)
    dic['bi'] = col(6, 7)
    dic['xyzd'] = col(7, 11).contiguous()
    dic['d'] = dic['xyzd'][:, 3:4].clone()
    dic['yaw'] = (col(11, 12), col(12, 13))
    return dic


def extract_outputs(outputs, tasks=()):
    """Raw network rows (m, 9|10) -> processed dictionary, or the raw per-task slices when `tasks` is
    given (reference process.py:231-278).  The dictionary tensors are detached CPU tensors."""
    slices = {'x': outputs[:, 0:1], 'y': outputs[:, 1:2], 'd': outputs[:, 2:4], 'h': outputs[:, 4:5],
              'w': outputs[:, 5:6], 'l': outputs[:, 6:7], 'ori': outputs[:, 7:9]}
    if outputs.shape[1] == 10:
        slices['aux'] = outputs[:, 9:10]
    if len(tasks) >= 1:
        assert isinstance(tasks, tuple), "tasks need to be a tuple"
        return [slices[task] for task in tasks]
    dev = engine._require_cuda(outputs.device if outputs.is_cuda else None)
    out, _ = engine.extract_outputs_device(outputs.detach().to(dev))
    return packed_to_dict(out, outputs.shape[1])


def extract_outputs_mono(outputs, tasks=None):
    """Legacy 'monoloco_p' outputs (m, 9) = x, y, z, log(b/z), h, w, l, sin, cos -> raw slices (tasks given) or the
    processed dictionary of detached CPU tensors (reference process.py:330-360)."""
    slices = {'xyz': outputs[:, 0:3], 'zb': outputs[:, 2:4], 'h': outputs[:, 4:5], 'w': outputs[:, 5:6],
              'l': outputs[:, 6:7], 'ori': outputs[:, 7:9]}
    if tasks is not None:
        assert isinstance(tasks, tuple), "tasks need to be a tuple"
        return [slices[task] for task in tasks]
    dev = engine._require_cuda(outputs.device if outputs.is_cuda else None)
    o = engine.extract_outputs_mono_device(outputs.detach().to(dev)).cpu()
    dic = {key: el.detach().cpu() for key, el in slices.items()}
    dic['xyzd'] = o[:, 0:4].clone()
    dic['d'], dic['bi'] = o[:, 3:4].clone(), o[:, 4:5].clone()
    dic['yaw'] = (o[:, 5:6].clone(), o[:, 6:7].clone())
    return dic


def laplace_sampling(outputs, n_samples):
    """n_samples draws of Laplace(mu = outputs[:,0], b = |outputs[:,1]|) per row -> (n_samples, m) on the device of
    `outputs` (reference process.py:101-122).  The reference re-seeds torch's generator with 1 on every call; here the
    library's counter-based generator is seeded with 1, so two calls give the same draws as well, but the streams
    differ from torch's (statistical parity)."""
    home = outputs.device
    xx = engine.laplace_sampling_device(outputs.detach().to(engine._require_cuda(home if home.type == 'cuda' else None)),
                                        n_samples, seed=1)
    return xx.to(home)


def extract_labels_aux(labels, tasks=None):
    """reference process.py:281-290."""
    dic = {'aux': labels[:, 0:1]}
    if tasks is not None:
        assert isinstance(tasks, tuple), "tasks need to be a tuple"
        return [dic[task] for task in tasks]
    return {key: el.detach().cpu() for key, el in dic.items()}


def extract_labels(labels, tasks=None):
    """Label columns theta, psi, z, d, h, w, l, sin, cos, yaw, aux (reference process.py:293-304)."""
    dic = {'x': labels[:, 0:1], 'y': labels[:, 1:2], 'z': labels[:, 2:3], 'd': labels[:, 3:4],
           'h': labels[:, 4:5], 'w': labels[:, 5:6], 'l': labels[:, 6:7], 'ori': labels[:, 7:9],
           'aux': labels[:, 10:11]}
    if tasks is not None:
        assert isinstance(tasks, tuple), "tasks need to be a tuple"
        return [dic[task] for task in tasks]
    return {key: el.detach().cpu() for key, el in dic.items()}


def cluster_outputs(outputs, clusters):
    """(ml*mr, C) -> (ml, mr, C) view; clusters == 0 means 'no right keypoints' (reference
    process.py:307-316)."""
    if clusters == 0:
        clusters = max(1, round(outputs.shape[0] / 2))
    assert outputs.shape[0] % clusters == 0, "Unexpected number of inputs"
    return outputs.view(-1, clusters, outputs.shape[1])


def filter_outputs(outputs):
    """Per left person keep the pair rows whose aux logit (last column) equals the row maximum; exact
    ties keep several rows (reference process.py:319-327).  Returns (rows, mask)."""
    val = outputs[:, :, -1]
    mask = val >= val.max(dim=1, keepdim=True).values  # index selection only, no path arithmetic
    return outputs[mask], mask
