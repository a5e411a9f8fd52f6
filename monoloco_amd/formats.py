"""On-disk formats either side of the hot path, on the native (C-ABI) reader / writer.

  * read_pifpaf_json / load_pifpaf -- an OpenPifPaf ``*.predictions.json`` file straight to arrays:
    json.load + preprocess_pifpaf (reference monoloco/network/process.py:155-218) in one native pass,
    no per-number Python objects; the keypoints come back ready for ``Loco.forward``.
  * save_txts -- the KITTI result files of reference monoloco/eval/generate_kitti.py:202-253.
  * write_monoloco_json -- ``<image>.monoloco.json`` of reference monoloco/predict.py:266-268.
"""
import ctypes
import json
import os

import numpy as np
import torch

from . import _lib
from .utils.camera import xyz_from_distance


def _fcheck(code):
    if code != 0:
        msg = _lib.load().ml_formats_last_error()
        raise _lib.MonolocoHipError("monoloco_hip formats error %d: %s" % (code, msg.decode() if msg else '?'))


def _dptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)) if a is not None else None


def parse_pifpaf_text(text, im_size=None, enlarge_boxes=True, min_conf=0.):
    """JSON text (bytes or str) of a PifPaf prediction list -> (boxes (m,5) float64 = x1,y1,x2,y2,conf,
    keypoints (m,3,17) float64), the arrays preprocess_pifpaf would return as nested lists."""
    lib = _lib.load()
    buf = text.encode('utf-8') if isinstance(text, str) else bytes(text)
    n = ctypes.c_int64(0)
    _fcheck(lib.ml_pifpaf_count(buf, len(buf), ctypes.byref(n)))
    boxes = np.empty((max(n.value, 1), 5), dtype=np.float64)
    kps = np.empty((max(n.value, 1), 3, 17), dtype=np.float64)
    m = ctypes.c_int64(0)
    w, h = (float(im_size[0]), float(im_size[1])) if im_size is not None else (0., 0.)
    _fcheck(lib.ml_pifpaf_parse(buf, len(buf), int(im_size is not None), w, h, int(bool(enlarge_boxes)),
                                float(min_conf), n.value, _dptr(boxes), _dptr(kps), ctypes.byref(m)))
    return boxes[:m.value], kps[:m.value]


def read_pifpaf_json(path, im_size=None, enlarge_boxes=True, min_conf=0., device=None):
    """File -> (boxes float64 ndarray (m,5), keypoints float32 tensor (m,3,17) on `device`)."""
    with open(path, 'rb') as f:
        boxes, kps = parse_pifpaf_text(f.read(), im_size, enlarge_boxes, min_conf)
    t = torch.from_numpy(kps.astype(np.float32))
    return boxes, (t.to(device) if device is not None else t)


def load_pifpaf(path, im_size=None, enlarge_boxes=True, min_conf=0.):
    """Drop-in for ``preprocess_pifpaf(json.load(open(path)), ...)``: nested Python lists."""
    with open(path, 'rb') as f:
        boxes, kps = parse_pifpaf_text(f.read(), im_size, enlarge_boxes, min_conf)
    return boxes.tolist(), kps.tolist()


def _col(x, m):
    """tensor / array / list of m scalars (or (m,1)) -> float64 (m,) exactly as float(x[idx]) would read it."""
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(m))


def kitti_txt(uv_boxes, xyz, bis, epis, alphas=None, rys=None, hwls=None, zzs_geom=None, tt=None, cat=None,
              conf_scale=0.035):
    """The text of one KITTI result file (one line per person), formatted natively."""
    lib = _lib.load()
    m = len(uv_boxes)
    if m == 0:
        return ""
    boxes = np.ascontiguousarray(np.asarray(uv_boxes, dtype=np.float64).reshape(m, 5))
    if isinstance(xyz, torch.Tensor):
        xyz = xyz.detach().cpu().numpy()
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(m, -1)[:, :3])
    arr = dict(bi=_col(bis, m), epi=_col(epis, m), cat=_col(cat, m),
               alpha=_col(alphas, m) if alphas is not None else None,
               ry=_col(rys, m) if rys is not None else None,
               zz=_col(zzs_geom, m) if zzs_geom is not None else None,
               tt=np.asarray(tt, dtype=np.float64).reshape(3) if tt is not None else None)
    hwl = None
    if hwls is not None:
        hwl = np.ascontiguousarray(np.stack([_col(c, m) for c in hwls], axis=1))
    cap = 64 + m * 24 * 340  # '%f ' of a double is at most 318 characters
    out = ctypes.create_string_buffer(cap)
    written = ctypes.c_int64(0)
    _fcheck(lib.ml_kitti_txt_format(m, _dptr(boxes), _dptr(xyz), _dptr(arr['bi']), _dptr(arr['epi']), _dptr(arr['alpha']),
                                    _dptr(arr['ry']), _dptr(hwl), _dptr(arr['zz']), _dptr(arr['tt']), _dptr(arr['cat']),
                                    float(conf_scale), out, cap, ctypes.byref(written)))
    return out.raw[:written.value].decode('ascii')


def save_txts(path_txt, all_inputs, all_outputs, all_params, net='monoloco', cat=None):
    """Same call as the reference's save_txts (generate_kitti.py:202-253): `all_inputs` = boxes of
    preprocess_pifpaf, `all_outputs` as assembled in GenerateKitti.run (:112-131), `all_params` = [kk, tt]."""
    assert net in ('monoloco', 'monstereo', 'geometric', 'baseline', 'monoloco_pp')
    alphas = rys = hwls = zzs = tt = None
    if net in ('monstereo', 'monoloco_pp'):
        xyzd, bis, epis, yaws, hs, ws, ls = all_outputs[:]
        xyz = xyzd[:, 0:3]
        alphas, rys, hwls = yaws[0], yaws[1], (hs, ws, ls)
        conf_scale = 0.035 if net == 'monoloco_pp' else 0.033
    elif net in ('monoloco', 'geometric'):
        dds, bis, epis, zzs_geom, xy_centers = all_outputs[:]
        xyz = xyz_from_distance(dds, xy_centers)
        zzs = zzs_geom if net == 'geometric' else None
        conf_scale = 0.05
    else:
        _, tt = all_params[:]
        xyz, bis, epis, zzs_geom, xy_centers = all_outputs[:]
        conf_scale = 0.05
    uv_boxes = all_inputs[:]
    assert len(uv_boxes) == len(list(xyz)), "Number of inputs different from number of outputs"
    text = kitti_txt(uv_boxes, xyz, bis, epis, alphas, rys, hwls, zzs, tt, cat, conf_scale)
    with open(path_txt, "w+") as ff:
        ff.write(text)


def write_monoloco_json(output_path, dic_out):
    """``<output_path>.monoloco.json`` with the post_process dictionary (reference predict.py:266-268)."""
    path = os.path.join(output_path + '.monoloco.json')
    with open(path, 'w') as ff:
        json.dump(dic_out, ff)
    return path
