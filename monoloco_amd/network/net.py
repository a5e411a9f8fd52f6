"""``Loco`` -- the inference API of the reference (monoloco/network/net.py:23-271) on the HIP engine.

Same constructor, ``forward`` and ``post_process`` signatures, same returned dictionaries (keys,
shapes, CPU fp32 tensors / Python lists, ``None`` for empty input).  Differences that are not
visible through the reference's own call sites: inputs may also be tensors (no list marshalling),
``device`` must be a HIP device (default: the current one) and ``post_process`` is vectorised (the
reference's per-person Python loop is quadratic in the number of persons).
"""
import ctypes
import logging
from collections import defaultdict
from itertools import chain as _chain_cls
from operator import itemgetter as _itemgetter

_chain = _chain_cls.from_iterable

import numpy as np
import torch

from .. import engine
from ..utils import xyz_from_distance
from ..utils.iou import get_iou_matches, get_iou_matches_ordered
from .architectures import LocoModel, MonolocoModel
from .process import extract_outputs_mono, packed_to_dict, unnormalize_bi


def _f64_list(t):
    """A column of fp32 results as Python floats (each the exact double of its fp32 value: what float(tensor[i]) yields)."""
    a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t, dtype=np.float32)
    return a.reshape(-1).astype(np.float64).tolist()


def _lists_to_f32(keypoints):
    """[m][3][17] nested Python lists (preprocess_pifpaf's keypoints) -> one (m, 3, 17) float32 array.  np.fromiter over the
    flattened lists is ~1.5x faster than np.asarray on nested lists (the per-frame host cost the reference's list API implies);
    anything that is not a plain list of [3][17] lists -- checked person by person -- takes np.asarray, which raises ValueError on
    ragged input like the reference's torch.tensor(keypoints)."""
    m = len(keypoints)
    # every person is checked: rows of 16 / 18 / 17 values would also sum to 51 per person and be reshaped silently
    if type(keypoints) is list and all(type(p) is list and len(p) == 3 and type(p[0]) is list
                                       and len(p[0]) == len(p[1]) == len(p[2]) == 17 for p in keypoints):
        it = _chain(_chain(keypoints))
        try:
            arr = np.fromiter(it, dtype=np.float32, count=m * 51)
            if next(it, None) is None:
                return arr.reshape(m, 3, 17)
        except (ValueError, TypeError):
            pass
    return np.asarray(keypoints, dtype=np.float32)


_PYHOST = [False]


def _pyhost():
    """The optional CPython helper (monoloco_amd/lib/libmonoloco_pyhost.so, csrc/pyhost.c) or None: nested keypoint lists -> the
    pinned staging buffer without an intermediate array.  Host-side marshalling only; numpy does the same job when it is missing."""
    if _PYHOST[0] is False:
        import os
        lib = None
        path = os.path.join(os.path.dirname(engine._lib.LIB_PATH), 'libmonoloco_pyhost.so')
        if os.path.exists(path):
            try:
                import sys
                lib = ctypes.PyDLL(path)
                lib.ml_py_version_hex.restype = ctypes.c_long
                if (lib.ml_py_version_hex() >> 16) != (sys.hexversion >> 16):   # built against another CPython (major.minor): numpy route
                    raise OSError("libmonoloco_pyhost.so was compiled for another Python")
                lib.ml_py_fill_kps.restype = ctypes.c_int
                lib.ml_py_fill_kps.argtypes = [ctypes.py_object, ctypes.c_void_p, ctypes.c_long]
            except (OSError, AttributeError):
                lib = None
        _PYHOST[0] = lib
    return _PYHOST[0]


# from this many matches on, the matched centres go through ml_xyz_from_distance on the device in one launch; below it through the
# library's host routine (ml_xyz_from_distance_host) on the (m,12) geometry block that is already on the host -- it was fetched for
# the dictionary; the reference's xy_centers are CPU tensors too.  Both are the same three correctly rounded fp32 operations per
# value in the reference's order: the same bits (tests/test_gpu_matching.py).
XYZ_REAL_DEVICE_MIN = 512


def _xyz_real(dds_real, xy_centers, idx_m):
    """[xyz_from_distance(dd, xy_centers[idx]).squeeze().tolist() for the matches] (reference net.py:242-247,
    camera.py:161-177): centre * fp32(dd) / sqrt(1 + x^2 + y^2) in fp32, as nested Python lists."""
    k = len(idx_m)
    if k >= XYZ_REAL_DEVICE_MIN:
        dd = torch.tensor(dds_real, dtype=torch.float64).to(torch.float32)   # (a Python float rounds to fp32 once, like torch.tensor(dd))
        return xyz_from_distance(dd, xy_centers[idx_m]).tolist()
    c = np.ascontiguousarray((xy_centers.numpy() if isinstance(xy_centers, torch.Tensor)
                              else np.asarray(xy_centers, dtype=np.float32))[idx_m], dtype=np.float32)
    dd = np.asarray(dds_real, dtype=np.float64).astype(np.float32)
    out = np.empty((k, 3), dtype=np.float32)
    engine.check(engine._lib.load().ml_xyz_from_distance_host(dd.ctypes.data, 0, c.ctypes.data, k, out.ctypes.data))
    return out.tolist()


class _LocoOut(dict):
    """The dictionary Loco.forward returns: a plain dict of the reference's keys, plus (as an attribute, not a key) the
    geometry block of the same keypoints that post_process would otherwise recompute."""
    __slots__ = ('_geo',)


class Loco:
    """Class for both MonoLoco++ (mode 'mono') and MonStereo (mode 'stereo')."""
    logger = logging.getLogger(__name__)
    LINEAR_SIZE_MONO = 256
    N_SAMPLES = 100
    _warned_default_device = False

    def __init__(self, model, mode, net=None, device=None, n_dropout=0, p_dropout=0.2, linear_size=1024):
        assert mode in ('mono', 'stereo'), "mode not recognized"
        self.mode = mode
        if net is None:
            self.net = 'monoloco_pp' if mode == 'mono' else 'monstereo'
        else:
            # The reference reads self.net before assigning it here (net.py:41) and raises AttributeError, so
            # net= is unusable there; the evident intent (net.py:47-60, 95-104) is implemented instead.
            assert net in ('monstereo', 'monoloco', 'monoloco_p', 'monoloco_pp')
            assert (net == 'monstereo') == (mode == 'stereo'), "Assert arguments mode and net are in conflict"
            self.net = net
        if self.net == 'monstereo':
            input_size, output_size = 68, 10
        elif self.net == 'monoloco_pp':
            input_size, output_size = 34, 9
        elif self.net == 'monoloco_p':  # legacy: MonolocoModel 34 -> 256 -> 9 (net.py:50-53)
            input_size, output_size, linear_size = 34, 9, 256
        else:  # legacy MonoLoco: 34 -> linear_size -> (d, log(b/d)), net.py:58-60
            input_size, output_size = 34, 2
        if device is None or torch.device(device).type != 'cuda':
            # the reference's default is the CPU (net.py:60-63: `device=None` -> torch.device('cpu')); this path has no CPU build
            if not torch.cuda.is_available() or device is not None:
                raise engine.MonolocoHipError(
                    "Loco(device=%r): the reference runs on the CPU here (monoloco/network/net.py:60-63, `device=None` means "
                    "torch.device('cpu')); monoloco_amd is the HIP path only and has no CPU fallback -- pass device=torch.device('cuda', i) "
                    "on a box with an AMD GPU%s" % (device, "" if torch.cuda.is_available() else " (no HIP device is visible to PyTorch)"))
            if not Loco._warned_default_device:
                Loco._warned_default_device = True
                self.logger.warning("Loco(device=None): the reference's default device is the CPU (net.py:60-63); monoloco_amd "
                                    "uses the current HIP device (cuda:%d) instead -- results come back as CPU tensors either way",
                                    torch.cuda.current_device())
        self.device = engine._require_cuda(device)
        self.n_dropout = n_dropout
        self.epistemic = bool(self.n_dropout > 0)
        if isinstance(model, str):
            if self.net in ('monoloco', 'monoloco_p'):
                self.model = MonolocoModel(p_dropout=p_dropout, input_size=input_size, linear_size=linear_size,
                                           output_size=output_size)
            else:
                self.model = LocoModel(p_dropout=p_dropout, input_size=input_size, output_size=output_size,
                                       linear_size=linear_size, device=self.device)
            self.model.load_state_dict(torch.load(model, map_location=lambda storage, loc: storage))
        else:
            self.model = model
        self.model.eval()
        sd = self.model.state_dict()
        assert sd['w1.weight'].shape[1] == input_size, "model input size does not match mode '%s'" % mode
        self.engine = engine.LocoEngine(sd, device=self.device,
                                        precision=getattr(self.model, 'precision', 'f16x2'),
                                        merge_w2w3=getattr(self.model, 'merge_w2w3', True))
        self._stage = {}   # per person count: pinned staging arrays + device buffers of the single-image path

    def forward(self, keypoints, kk, keypoints_r=None):
        """Pre-process, network forward and output extraction for one image (reference net.py:83-133).
        Returns the reference's dictionary of CPU tensors, or None when there are no keypoints."""
        if keypoints is None or len(keypoints) == 0:
            return None
        dev = self.device
        kk_list = kk.tolist() if hasattr(kk, 'tolist') else kk
        if self.net == 'monstereo' and not (isinstance(keypoints, torch.Tensor) and keypoints.is_cuda) \
                and not (isinstance(keypoints_r, torch.Tensor) and keypoints_r.is_cuda):
            # one stereo image pair from the host (the reference's call, predict.py:244-245): ONE foreign call, pinned buffers both ways
            staged = self._forward_stereo_staged(keypoints, keypoints_r, engine.inverse_intrinsics(kk_list))
            if staged is not None:   # (None: exact ties of the aux logit -- the rare frame takes the general route below)
                dic_out, geo_host = staged
                dic_out['epi'] = [0.] * len(keypoints)
                dic_out = _LocoOut(dic_out)
                dic_out._geo = (keypoints, kk_list, dic_out['d'], geo_host)
                return dic_out
        if self.net == 'monoloco_pp' and not (isinstance(keypoints, torch.Tensor) and keypoints.is_cuda):
            # one image from the host (the reference's call, predict.py:231-249): lists -> ONE float32 array, staged below
            # (a plain list stays a list here: _forward_pp_staged walks it straight into the pinned staging buffer)
            kps = keypoints if type(keypoints) is list else (
                _lists_to_f32(keypoints) if not isinstance(keypoints, torch.Tensor) else keypoints.float().numpy())
        else:
            kps = engine._dev_f32(keypoints, dev)
        kinv = engine.inverse_intrinsics(kk_list)
        geo_host = None
        if self.net == 'monoloco':
            # legacy MonoLoco (net.py:95-100): zero-centred inputs, outputs (d, log(b/d))
            x = engine.preprocess_mono(kps, kk.tolist() if isinstance(kk, torch.Tensor) else kk, device=dev,
                                       zero_center=True)
            raw = self.engine.forward_raw(x)
            dic_out = {'d': raw[:, 0:1].cpu(), 'bi': unnormalize_bi(raw).cpu()}
            n_out = kps.shape[0]
        elif self.net == 'monoloco_p':
            # legacy (net.py:102-104): plain inputs, MonolocoModel with 9 outputs, extract_outputs_mono
            x = engine.preprocess_mono(kps, kk.tolist() if isinstance(kk, torch.Tensor) else kk, device=dev)
            dic_out = extract_outputs_mono(self.engine.forward_raw(x))
            n_out = kps.shape[0]
        elif self.net == 'monoloco_pp':
            dic_out, geo_host, kps = self._forward_pp_staged(kps, kinv)   # (kps: now the float32 array / device tensor)
            n_out = kps.shape[0]
        else:
            if keypoints_r is not None and len(keypoints_r) > 0:
                kps_r = engine._dev_f32(keypoints_r, dev)
            else:
                kps_r = kps[0:1].clone()  # reference net.py:115-116
            buf, out, geo = self._packed_buffers(kps.shape[0])
            res = self.engine.forward_stereo(kps, kps_r, kinv, want_raw_all=True, out=out)
            if int(res['ties'].item()) == 0:
                dic_out, geo_host = self._fetch_with_geometry(buf, out, geo, kps, kinv, 10)
            else:
                # exact ties of the aux logit: the reference keeps every tied pair row (process.py:325)
                rows = engine.stereo_tied_rows(res['raw_all'], kps.shape[0], kps_r.shape[0])   # filter_outputs' mask, on the device
                out, _ = engine.extract_outputs_device(res['raw_all'], row_index=rows)
                dic_out = packed_to_dict(out, 10)
            n_out = kps.shape[0]
        if self.n_dropout > 0 and self.net != 'monstereo':
            # combined aleatoric + epistemic spread by MC-dropout (reference net.py:126-128,135-161): a tensor
            dic_out['epi'] = self.engine.epistemic_mono(kps, kinv, self.n_dropout,
                                                        p_dropout=getattr(self.model, 'p_dropout', 0.2),
                                                        n_samples=self.N_SAMPLES).cpu()   # == self.epistemic_uncertainty(inputs)
        else:
            dic_out['epi'] = [0.] * n_out
        if geo_host is not None:
            # the geometry block post_process needs, computed behind the network from the keypoints that were on the device
            # anyway and fetched in the same copy; post_process uses it when it is handed these very objects again (identity +
            # length checks only: a caller that edits the keypoints list or dic['d'] IN PLACE between forward and post_process
            # must drop the cache -- `dic_out._geo = None` -- or pass copies; the reference's call sites do neither)
            dic_out = _LocoOut(dic_out)
            dic_out._geo = (keypoints, kk_list, dic_out['d'], geo_host)
        return dic_out

    def epistemic_uncertainty(self, inputs):
        """Apply dropout at test time to obtain combined aleatoric + epistemic uncertainty (reference net.py:135-161).
        `inputs`: the PRE-PROCESSED (m, 34) network inputs (preprocess_monoloco's result); returns the (m,) standard deviation
        over n_dropout stochastic passes x N_SAMPLES Laplace draws, on self.device like the reference's."""
        assert self.net in ('monoloco', 'monoloco_p', 'monoloco_pp'), "Not supported for MonStereo"
        assert self.n_dropout > 0, "n_dropout must be positive (the reference returns the std of an empty tensor: nan)"
        return self.engine.epistemic_inputs(inputs, self.n_dropout, p_dropout=getattr(self.model, 'p_dropout', 0.2),
                                            n_samples=self.N_SAMPLES)

    def _forward_pp_staged(self, kps, kinv):
        """MonoLoco++ forward of one image with every buffer cached per person count: the keypoints go through a pinned staging
        array (one asynchronous H2D), the network result (m,16) + the post-process geometry (m,12) come back in one
        asynchronous D2H into a pinned array behind the launches; the host only waits once, for that copy.  Returns
        (dictionary of fresh CPU tensors, fresh (m,12) geometry tensor)."""
        lib = engine._lib.load()
        dev = self.device
        m = len(kps) if type(kps) is list else int(kps.shape[0])
        stride = engine._lib.ML_OUT_STRIDE
        st = self._stage.get(m)
        if st is None:
            if len(self._stage) >= 16:
                # the pinned staging buffers go back to torch's host allocator: the handle must stop trusting their addresses
                # (the pinned-buffer contract of ml_loco_frame_*, include/monoloco_hip.h)
                engine._lib.load().ml_loco_forget_pinned(self.engine._h, None)
                self._stage.clear()
            pin_in = torch.empty((m, 3, 17), dtype=torch.float32).pin_memory()
            pin_out = torch.empty((m * (stride + 12),), dtype=torch.float32).pin_memory()
            st = dict(pin_in=pin_in, np_in=pin_in.numpy(), dev_in=torch.empty((m, 3, 17), dtype=torch.float32, device=dev),
                      buf=torch.empty((m * (stride + 12),), dtype=torch.float32, device=dev),
                      xyzds=torch.empty((m, engine._lib.ML_XYZDS_STRIDE), dtype=torch.float32, device=dev),
                      pin_out=pin_out, np_out=pin_out.numpy())
            # where every value of the result layout lies in the pinned [packed (m,16) | geometry (m,12)] block
            idx = np.arange(m * stride).reshape(m, stride)
            st['perm'] = np.concatenate([idx[:, [8, 9, 10, 4, 5, 6, 3]].T.reshape(-1), idx[:, 12:14].reshape(-1), idx[:, 0:4].reshape(-1),
                                         np.arange(m * stride, m * (stride + 12))]).astype(np.intp)
            st['sizes'] = [7 * m, 2 * m, 4 * m, 12 * m]
            st['p_in'] = ctypes.c_void_p(st['dev_in'].data_ptr())
            st['p_out'] = ctypes.c_void_p(st['buf'].data_ptr())
            st['p_d'] = ctypes.c_void_p(st['buf'].data_ptr() + 3 * 4)
            st['p_geo'] = ctypes.c_void_p(st['buf'].data_ptr() + m * stride * 4)
            st['p_xyzds'] = ctypes.c_void_p(st['xyzds'].data_ptr())
            st['p_pin_in'] = ctypes.c_void_p(pin_in.data_ptr())
            st['p_pin_out'] = ctypes.c_void_p(pin_out.data_ptr())
            self._stage[m] = st
        stream = engine._stream(dev)
        kinv_p = engine.kinv_ptr(kinv)
        if isinstance(kps, torch.Tensor):     # already on the device: no staging
            assert tuple(kps.shape[1:]) == (3, 17), "keypoints must be (m, 3, 17)"
            p_in = ctypes.c_void_p(kps.data_ptr())
            with torch.cuda.device(dev):
                engine.check(lib.ml_loco_forward_mono(self.engine._h, p_in, m, kinv_p, None, None, st['p_out'], st['p_xyzds'], stream))
                engine.check(lib.ml_post_geometry_strided(p_in, m, kinv_p, st['p_d'], stride, st['p_geo'], stream))
                st['pin_out'].copy_(st['buf'], non_blocking=True)
                torch.cuda.current_stream(dev).synchronize()
        else:                                 # the frame in ONE foreign call: H2D, pipeline, geometry, D2H, stream sync
            if type(kps) is list:   # nested lists: the C walk (csrc/pyhost.c) where it applies, numpy otherwise
                ph = _pyhost()
                if ph is None or ph.ml_py_fill_kps(kps, st['p_pin_in'], m) != 0:
                    np.copyto(st['np_in'], _lists_to_f32(kps))
                kps = st['np_in']
            else:
                assert kps.shape[1:] == (3, 17), "keypoints must be (m, 3, 17)"
                np.copyto(st['np_in'], kps)
            if torch.cuda.current_device() == dev.index:   # (the usual case: no device switch, no context manager around the call)
                engine.check(lib.ml_loco_frame_mono(self.engine._h, st['p_pin_in'], m, kinv_p, st['p_in'], st['p_out'], st['p_xyzds'],
                                                    st['p_pin_out'], stream))
            else:
                with torch.cuda.device(dev):
                    engine.check(lib.ml_loco_frame_mono(self.engine._h, st['p_pin_in'], m, kinv_p, st['p_in'], st['p_out'],
                                                        st['p_xyzds'], st['p_pin_out'], stream))
        # the reference's dictionary (process.py:240-278) out of ONE fresh buffer: 7 single columns (h w l bi yaw yaw_ego d), then
        # ori (m,2), xyzd (m,4) and the (m,12) geometry block -- one gather of the pinned result through an index vector cached per
        # person count, one torch.from_numpy, every output a view of its own range
        t = torch.from_numpy(st['np_out'].take(st['perm']))
        cols, ori, xyzd, geo = t.split_with_sizes(st['sizes'])   # (one call instead of four slices)
        h_, w_, l_, bi_, yaw_, yawe_, d_ = cols.view(7, m, 1).unbind(0)
        dic = {'h': h_, 'w': w_, 'l': l_, 'ori': ori.view(m, 2), 'bi': bi_, 'xyzd': xyzd.view(m, 4), 'd': d_, 'yaw': (yaw_, yawe_)}
        return dic, geo.view(m, 12), kps

    def _forward_stereo_staged(self, keypoints, keypoints_r, kinv):
        """MonStereo forward of one image pair with every buffer cached per (left, right) person count: both keypoint sets go
        through pinned staging arrays the pre-process kernels read directly, the per-left winners' packed rows (ml,16), the
        post-process geometry (ml,12), the tie count and the arg-max indices come back in one pinned block the last launch
        completes (ml_loco_frame_stereo).  Returns (dictionary of fresh CPU tensors, fresh (ml,12) geometry tensor), or None when
        some left person's best aux logit is tied (the reference keeps every tied pair row: the general route handles it)."""
        lib = engine._lib.load()
        dev = self.device
        if keypoints_r is None or len(keypoints_r) == 0:
            keypoints_r = keypoints[0:1]   # reference net.py:115-116: the first left pose stands in
        ml, mr = len(keypoints), len(keypoints_r)
        stride = engine._lib.ML_OUT_STRIDE
        key = ('stereo', ml, mr)
        st = self._stage.get(key)
        if st is None:
            if len(self._stage) >= 16:
                # the pinned staging buffers go back to torch's host allocator: the handle must stop trusting their addresses
                # (the pinned-buffer contract of ml_loco_frame_*, include/monoloco_hip.h)
                engine._lib.load().ml_loco_forget_pinned(self.engine._h, None)
                self._stage.clear()
            words = ml * (stride + 12) + 1 + ml
            pin_l = torch.empty((ml, 3, 17), dtype=torch.float32).pin_memory()
            pin_r = torch.empty((mr, 3, 17), dtype=torch.float32).pin_memory()
            pin_out = torch.empty((words,), dtype=torch.float32).pin_memory()
            st = dict(pin_l=pin_l, pin_r=pin_r, pin_out=pin_out, np_l=pin_l.numpy(), np_r=pin_r.numpy(), np_out=pin_out.numpy(),
                      dev_in=torch.empty(((ml + mr) * 51,), dtype=torch.float32, device=dev),
                      buf=torch.empty((words,), dtype=torch.float32, device=dev),
                      xyzds=torch.empty((ml, engine._lib.ML_XYZDS_STRIDE), dtype=torch.float32, device=dev))
            st['np_ties'] = st['np_out'][ml * (stride + 12):ml * (stride + 12) + 1].view(np.int32)
            idx = np.arange(ml * stride).reshape(ml, stride)
            # h w l bi yaw yaw_ego d aux as single columns, then ori (ml,2), xyzd (ml,4), the (ml,12) geometry block
            st['perm'] = np.concatenate([idx[:, [8, 9, 10, 4, 5, 6, 3, 7]].T.reshape(-1), idx[:, 12:14].reshape(-1), idx[:, 0:4].reshape(-1),
                                         np.arange(ml * stride, ml * (stride + 12))]).astype(np.intp)
            st['sizes'] = [8 * ml, 2 * ml, 4 * ml, 12 * ml]
            for name, t in (('p_l', pin_l), ('p_r', pin_r), ('p_out', pin_out), ('p_in', st['dev_in']), ('p_buf', st['buf']),
                            ('p_xyzds', st['xyzds'])):
                st[name] = ctypes.c_void_p(t.data_ptr())
            self._stage[key] = st
        ph = _pyhost()
        for src, n, np_dst, p_dst in ((keypoints, ml, st['np_l'], st['p_l']), (keypoints_r, mr, st['np_r'], st['p_r'])):
            if type(src) is list:
                if ph is None or ph.ml_py_fill_kps(src, p_dst, n) != 0:
                    np.copyto(np_dst, _lists_to_f32(src))
            else:
                arr = src.float().numpy() if isinstance(src, torch.Tensor) else np.asarray(src, dtype=np.float32)
                assert arr.shape[1:] == (3, 17), "keypoints must be (m, 3, 17)"
                np.copyto(np_dst, arr)
        kinv_p = engine.kinv_ptr(kinv)
        stream = engine._stream(dev)
        call = lambda: engine.check(lib.ml_loco_frame_stereo(self.engine._h, st['p_l'], ml, st['p_r'], mr, kinv_p, st['p_in'], st['p_buf'],
                                                             st['p_xyzds'], st['p_out'], stream))
        if torch.cuda.current_device() == dev.index:
            call()
        else:
            with torch.cuda.device(dev):
                call()
        if int(st['np_ties'][0]) != 0:
            return None
        t = torch.from_numpy(st['np_out'].take(st['perm']))
        cols, ori, xyzd, geo = t.split_with_sizes(st['sizes'])
        h_, w_, l_, bi_, yaw_, yawe_, d_, aux_ = cols.view(8, ml, 1).unbind(0)
        dic = {'h': h_, 'w': w_, 'l': l_, 'ori': ori.view(ml, 2), 'aux': aux_, 'bi': bi_, 'xyzd': xyzd.view(ml, 4), 'd': d_,
               'yaw': (yaw_, yawe_)}
        return dic, geo.view(ml, 12)

    def _packed_buffers(self, m):
        """One device allocation for the packed (m,16) network result and the (m,12) post-process geometry."""
        buf = torch.empty((m * (engine._lib.ML_OUT_STRIDE + 12),), dtype=torch.float32, device=self.device)
        n = m * engine._lib.ML_OUT_STRIDE
        return buf, buf[:n].view(m, engine._lib.ML_OUT_STRIDE), buf[n:].view(m, 12)

    def _fetch_with_geometry(self, buf, out, geo, kps, kinv, n_cols):
        """post_geometry on the device from the packed distances (column 3), then ONE copy of both blocks off the device."""
        m = out.shape[0]
        # (engine.post_geometry without its argument checks: everything here was just produced by this class)
        with torch.cuda.device(self.device):
            engine.check(engine._lib.load().ml_post_geometry_strided(
                engine._ptr(kps), m, engine.fptr(kinv), ctypes.c_void_p(out.data_ptr() + 3 * 4), engine._lib.ML_OUT_STRIDE,
                engine._ptr(geo), engine._stream(self.device)))
        n = m * engine._lib.ML_OUT_STRIDE
        host = buf.cpu()
        return packed_to_dict(host[:n].view(m, engine._lib.ML_OUT_STRIDE), n_cols), host[n:].view(m, 12)

    @staticmethod
    def post_process(dic_in, boxes, keypoints, kk, dic_gt=None, iou_min=0.3, reorder=True, verbose=False):
        """Final per-image dictionary for visualisation / json (reference net.py:163-248): optional
        IoU matching with ground truth, representative pixels, back-projected xyz and confidence."""
        dic_out = defaultdict(list)
        if dic_in is None:
            return dic_out
        # ground-truth association (reference net.py:170-192): matched detections first (left to right when
        # `reorder`), then the unmatched ones in input order
        matches, boxes_gt, dds_gt = [], [], []
        if dic_gt:
            boxes_gt = dic_gt['boxes']
            dds_gt = [ys[3] for ys in dic_gt['ys']]
            # all m x g IoUs, the greedy pass by confidence and (when `reorder`) the left-to-right order in one native call
            matches = (get_iou_matches_ordered if reorder else get_iou_matches)(boxes, boxes_gt, iou_min=iou_min)
        if verbose:
            print("found {} matches with ground-truth".format(len(matches)) if dic_gt else "NO ground-truth associated")
        if matches:
            taken = {pair[0] for pair in matches}
            not_matches = [idx for idx in range(len(boxes)) if idx not in taken]
        else:
            not_matches = list(range(len(boxes)))
        all_idxs = [pair[0] for pair in matches] + not_matches
        dic_out['gt'] = [True] * len(matches) + [False] * len(not_matches)

        # device geometry, whole image in ONE launch and one copy back (ml_post_geometry): representative pixels,
        # normalised centre, back-projected xyz
        d_all = torch.as_tensor(dic_in['d'], dtype=torch.float32).reshape(-1)
        kk_list = kk.tolist() if hasattr(kk, 'tolist') else kk
        cached = getattr(dic_in, '_geo', None)
        if cached is not None and cached[0] is keypoints and cached[2] is dic_in['d'] and cached[1] == kk_list \
                and cached[3].shape[0] == len(keypoints) == d_all.shape[0]:
            geo = cached[3]   # computed by forward() on these keypoints / intrinsics / distances
            n_pred = d_all.shape[0]
        else:
            kps_t = keypoints if isinstance(keypoints, torch.Tensor) else torch.tensor(keypoints, dtype=torch.float32)
            m_kp = kps_t.shape[0]
            n_pred = min(d_all.shape[0], m_kp)
            d_fit = torch.zeros(m_kp, dtype=torch.float32)
            d_fit[:n_pred] = d_all[:n_pred]
            geo = engine.post_geometry(kps_t, kk_list, d_fit).cpu()
        xy_centers = geo[:, 6:9]
        g64 = geo.numpy().astype(np.float64)
        xyz_all = g64[:n_pred, 9:12]
        dist_all = np.sqrt(xyz_all[:, 0] ** 2 + xyz_all[:, 1] ** 2 + xyz_all[:, 2] ** 2)
        uv = np.rint(g64[:, 0:6]).astype(int)
        uv_s, uv_h, uv_c = uv[:, 0:2], uv[:, 2:4], uv[:, 4:6]
        has_yaw = 'yaw' in dic_in
        has_aux = 'aux' in dic_in
        if not has_yaw and all_idxs:
            dic_out['angles']  # the reference touches the key before the KeyError (net.py:231)
        # assemble the output column by column (the reference appends person by person, net.py:206-240); key creation
        # order is kept because the dictionary is dumped to json as it is
        if all_idxs:
            # every per-person scalar comes out of ONE tolist() per column (a Python float of an fp32 value is that value as a
            # double, exactly what float(tensor[i]) gives the reference); no per-person tensor indexing
            d_l = _f64_list(d_all)
            bi_l = _f64_list(dic_in['bi'])
            dist_l = dist_all.tolist()
            epi = dic_in['epi']
            epi_l = _f64_list(epi) if isinstance(epi, torch.Tensor) else [float(e) for e in epi]
            identity = all_idxs == list(range(len(all_idxs)))   # no ground truth: input order
            if identity:
                pick = lambda col: col[:len(all_idxs)]
            elif len(all_idxs) > 1:   # matched first: one C-level gather per column (operator.itemgetter), no Python loop
                gather = _itemgetter(*all_idxs)
                pick = lambda col: list(gather(col))
            else:
                pick = lambda col: [col[i] for i in all_idxs]
            columns = [
                ('boxes', pick(list(boxes))),
                ('confs', [0.035 * (boxes[i][-1]) / (bi_l[i] / dist_l[i]) for i in all_idxs]),
                ('dds_pred', pick(d_l)),
                ('stds_ale', pick(bi_l)),
                ('stds_epi', pick(epi_l)),
                ('xyz_pred', pick(xyz_all.tolist())),
                ('uv_kps', pick(list(keypoints))),
                ('uv_centers', pick(uv_c.tolist())),
                ('uv_shoulders', pick(uv_s.tolist())),
                ('uv_heads', pick(uv_h.tolist())),
            ]
            if has_yaw:
                yaw_pred, yaw_ego = dic_in['yaw']
                columns.append(('angles', pick(_f64_list(yaw_pred))))
                columns.append(('angles_egocentric', pick(_f64_list(yaw_ego))))
                # mono: the 'aux' key exists and stays empty (reference net.py:237-240)
                columns.append(('aux', pick(_f64_list(dic_in['aux'])) if has_aux else []))
            for key, values in columns:
                dic_out[key] = values
        if matches:
            # the ground-truth side of every match at once (the reference calls xyz_from_distance per match, net.py:242-247)
            idx_m = [pair[0] for pair in matches]
            dds_real = [dds_gt[pair[1]] for pair in matches]
            dic_out['dds_real'] = dds_real
            dic_out['boxes_gt'] = [boxes_gt[pair[1]] for pair in matches]
            dic_out['xyz_real'] = _xyz_real(dds_real, xy_centers, idx_m)
        return dic_out

    @staticmethod
    def social_distance(dic_out, args):
        """Per person: does it violate social distancing / stand in an F-formation (reference net.py:250-264)."""
        from ..activity import social_interactions
        xz = [[xx[0], xx[2]] for xx in dic_out['xyz_pred']]
        dic_out['social_distance'] = [bool(social_interactions(idx, xz, dic_out['angles'], dic_out['dds_pred'],
                                                               stds=dic_out['stds_ale'],
                                                               threshold_prob=args.threshold_prob,
                                                               threshold_dist=args.threshold_dist, radii=args.radii))
                                      for idx, _ in enumerate(dic_out['xyz_pred'])]
        return dic_out

    @staticmethod
    def raising_hand(dic_out, keypoints):
        """Per person 'left' / 'right' / 'both' / None (reference net.py:267-270)."""
        from ..activity import is_raising_hand
        dic_out['raising_hand'] = [is_raising_hand(keypoint) for keypoint in keypoints]
        return dic_out
