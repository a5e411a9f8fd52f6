"""Camera geometry with the reference's call signatures (reference monoloco/utils/camera.py),
executed by the stand-alone HIP kernels of ``csrc/geom_ops.h`` through the C ABI.

Each function accepts what the reference accepts (lists, numpy arrays, torch tensors) and returns
a torch tensor on the device of its input: host inputs give CPU tensors (computed on the GPU and
copied back), device inputs stay on the device.  There is no host arithmetic path for tensors (the one-point
Python-list form of ``to_cartesian`` is Python doubles in the reference and here).
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from .._lib import check, fptr
from ..engine import _dev_f32, _ptr, _require_cuda, _stream, inverse_intrinsics

_MODES = {'center': 0, 'bottom': 1, 'head': 2, 'shoulder': 3, 'hip': 4, 'ankle': 5}


def _home(x):
    """(device the result should live on, device to compute on)."""
    if isinstance(x, torch.Tensor) and x.is_cuda:
        return x.device, x.device
    return torch.device('cpu'), _require_cuda(None)


def pixel_to_camera(uv_tensor, kk, z_met):
    """reference camera.py:10-29 -- [u, v, 1] . inverse(K)^T * z_met for (m,2), (m,x,2) or (m,2,x)."""
    home, dev = _home(uv_tensor)
    uv = _dev_f32(uv_tensor, dev)
    squeeze = uv.dim() == 1
    if squeeze:
        uv = uv.unsqueeze(0)
    if uv.shape[-1] != 2:
        uv = uv.permute(0, 2, 1)
        assert uv.shape[-1] == 2, "Tensor size not recognized"
    uv = uv.contiguous()
    n = uv.numel() // 2
    out = torch.empty(uv.shape[:-1] + (3,), dtype=torch.float32, device=dev)
    kinv = inverse_intrinsics(kk.detach().cpu().numpy() if isinstance(kk, torch.Tensor) else kk)
    with torch.cuda.device(dev):
        check(_lib.load().ml_pixel_to_camera(_ptr(uv), n, fptr(kinv), float(z_met), _ptr(out), _stream(dev)))
    if squeeze:
        out = out.reshape(3)  # reference: F.pad of a 1-D [u, v] then matmul with (3,3) -> shape (3,)
    return out.to(home)


def get_keypoints(keypoints, mode):
    """reference camera.py:69-107 -- (m,3,17) or (3,17) -> (m,2)."""
    assert mode in _MODES
    home, dev = _home(keypoints)
    kps = _dev_f32(keypoints, dev)
    if kps.dim() == 2:
        kps = kps.unsqueeze(0)
    assert kps.dim() == 3 and kps.shape[1] == 3, "tensor dimensions not recognized"
    assert kps.shape[2] == 17, "17 COCO keypoints expected"
    m = kps.shape[0]
    out = torch.empty((m, 2), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_get_keypoints(_ptr(kps), m, _MODES[mode], _ptr(out), _stream(dev)))
    return out.to(home)


def xyz_from_distance(distances, xy_centers):
    """reference camera.py:161-177 -- xy_centers * d / sqrt(1 + x^2 + y^2)."""
    home, dev = _home(xy_centers)
    scalar = isinstance(distances, (float, int))
    d = torch.tensor([float(distances)]) if scalar else distances
    d = _dev_f32(d, dev)
    c = _dev_f32(xy_centers, dev)
    if c.dim() == 1:
        c = c.unsqueeze(0)
    if d.dim() == 2:
        assert d.shape[-1] == 1, "Size of tensor not recognized"
        d = d.reshape(-1)
    assert c.shape[-1] == 3, "Size of tensor not recognized"
    m = c.shape[0]
    one = d.numel() == 1 and m >= 1
    assert one or d.numel() == m
    out = torch.empty((m, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_xyz_from_distance(_ptr(d), int(one), _ptr(c), m, _ptr(out), _stream(dev)))
    return out.to(home)


def to_cartesian(rtp, mode=None):
    """reference camera.py:223-248.  Tensor input: mode 'x'/'y' take rows (theta, psi, r) and return
    (m,1); any other mode takes rows (r, theta, psi) and returns (m,3).  A plain [r, theta, psi] list
    returns a python list like the reference."""
    if not isinstance(rtp, torch.Tensor):
        # the reference's list branch (camera.py:245-248) is three Python-double expressions on one point (dataset
        # preparation labels); same expressions, same doubles -- not a device path
        r, t, p = float(rtp[0]), float(rtp[1]), float(rtp[2])
        return [r * math.sin(p) * math.cos(t), r * math.cos(p), r * math.sin(p) * math.sin(t)]
    home, dev = _home(rtp)
    t = _dev_f32(rtp, dev)
    m = t.shape[0]
    code = {'x': 0, 'y': 1}.get(mode, 2)
    out = torch.empty((m, 1) if code < 2 else (m, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_to_cartesian(_ptr(t[:, 0:3].contiguous()), m, code, _ptr(out), _stream(dev)))
    return out.to(home)


def back_correct_angles(yaws, xyz):
    """reference camera.py:202-208 -- yaw + atan2(x, z), wrapped once into (-pi, pi]; returns (m,1)."""
    home, dev = _home(yaws)
    y = _dev_f32(yaws, dev).reshape(-1)
    p = _dev_f32(xyz, dev)
    m = y.shape[0]
    out = torch.empty((m,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_back_correct_angles(_ptr(y), _ptr(p[:, 0:3].contiguous()), m, _ptr(out), _stream(dev)))
    return out.view(-1, 1).to(home)
