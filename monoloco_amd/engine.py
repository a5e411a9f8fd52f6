"""Device engine: torch tensors in, C-ABI calls out.

PyTorch is used here for plumbing only -- device memory (``tensor.data_ptr()``), the current
HIP stream and host<->device copies.  All arithmetic of the hot path happens inside
``libmonoloco_hip.so`` (hand-written gfx950 kernels); nothing here falls back to torch ops.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import MonolocoHipError, check, fptr

PRECISIONS = {'f16x2': _lib.ML_PREC_F16X2, 'f16': _lib.ML_PREC_F16, 'bf16': _lib.ML_PREC_BF16}


def _require_cuda(device):
    if not torch.cuda.is_available():
        raise MonolocoHipError("no HIP device visible to PyTorch: the monoloco_amd hot path needs an AMD GPU "
                               "(there is no CPU fallback)")
    dev = torch.device(device if device is not None else 'cuda')
    if dev.type != 'cuda':
        raise MonolocoHipError("monoloco_amd runs on HIP devices only, got device %r" % (device,))
    if dev.index is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    return dev


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(dev):
    """The raw hipStream_t of torch's current stream on dev.  torch._C._cuda_getCurrentRawStream (what torch's own compiled-code
    launchers call) answers in ~0.3 us; torch.cuda.current_stream(dev).cuda_stream builds a Stream object first (~5 us per frame)."""
    if _RAW_STREAM is not None and dev.index is not None:
        return ctypes.c_void_p(_RAW_STREAM(dev.index))
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _dev_f32(x, dev, shape=None):
    """list / numpy / tensor -> contiguous fp32 tensor on dev (no copy if already there)."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32))
    x = x.to(device=dev, dtype=torch.float32).contiguous()
    if shape is not None:
        x = x.reshape(shape)
    return x


_KINV_CACHE = {}
_KINV_PTR = {}   # id(cached inverse) -> its ctypes float pointer (numpy's ctypes.data_as costs ~10 us per call)


def inverse_intrinsics(kk):
    """inverse(K) as 9 fp32 numbers, taken the way the reference takes it: torch.inverse on an
    fp32 CPU tensor (reference monoloco/utils/camera.py:23).  A camera's K is the same frame after frame:
    the last few inverses are kept (the returned array must not be modified)."""
    k32 = np.asarray(kk, dtype=np.float32).reshape(9)
    key = k32.tobytes()
    hit = _KINV_CACHE.get(key)
    if hit is None:
        if len(_KINV_CACHE) >= 64:
            _KINV_CACHE.clear()
            _KINV_PTR.clear()
        k = torch.as_tensor(k32.copy()).reshape(3, 3)
        hit = np.ascontiguousarray(torch.inverse(k).numpy().reshape(9).astype(np.float32))
        hit.setflags(write=False)
        _KINV_CACHE[key] = hit
        _KINV_PTR[id(hit)] = fptr(hit)
    return hit


def kinv_ptr(kinv):
    """ctypes pointer to the 9 floats of an inverse_intrinsics() result (cached with the array), or of any fp32 array."""
    p = _KINV_PTR.get(id(kinv))
    return p if p is not None else fptr(kinv)


# ------------------------------------------------------------------ stand-alone kernels
def preprocess_mono(kps, kk, z_met=10.0, device=None, want_centre=False, zero_center=False):
    """(m,3,17) pixel keypoints -> (m,34) normalised inputs [and (m,2) centres] on the device."""
    lib = _lib.load()
    dev = _require_cuda(device if device is not None else (kps.device if isinstance(kps, torch.Tensor) and kps.is_cuda else None))
    kps = _dev_f32(kps, dev)
    assert kps.dim() == 3 and kps.shape[1] == 3 and kps.shape[2] == 17, "keypoints must be (m, 3, 17)"
    m = kps.shape[0]
    x = torch.empty((m, 34), dtype=torch.float32, device=dev)
    c = torch.empty((m, 2), dtype=torch.float32, device=dev) if want_centre else None
    kinv = inverse_intrinsics(kk)
    with torch.cuda.device(dev):
        check(lib.ml_preprocess_mono(_ptr(kps), m, fptr(kinv), float(z_met), int(bool(zero_center)), _ptr(x), _ptr(c),
                                     _stream(dev)))
    return (x, c) if want_centre else x


def preprocess_rows(kps, kks, k_index, kps_r=None, device=None, z_met=10.0):
    """Dataset-preparation inputs in one launch: row i of kps (m,3,17) is normalised with kks[k_index[i]]
    (reference prep/preprocess_kitti.py:190-253).  kps_r (m,3,17) given -> (m,68) stereo rows [L, L - R]."""
    lib = _lib.load()
    dev = _require_cuda(device)
    kps = _dev_f32(kps, dev)
    assert kps.dim() == 3 and kps.shape[1] == 3 and kps.shape[2] == 17, "keypoints must be (m, 3, 17)"
    m = kps.shape[0]
    if kps_r is not None:
        kps_r = _dev_f32(kps_r, dev)
        assert kps_r.shape == kps.shape, "left and right keypoints must pair row by row"
    table = np.ascontiguousarray(np.stack([inverse_intrinsics(kk) for kk in kks]).astype(np.float32))
    idx = torch.as_tensor(np.asarray(k_index), dtype=torch.int32).to(dev).contiguous()
    assert idx.shape == (m,) and (m == 0 or (int(idx.min()) >= 0 and int(idx.max()) < len(table)))
    x = torch.empty((m, 68 if kps_r is not None else 34), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.ml_preprocess_rows(_ptr(kps), _ptr(kps_r), m, fptr(table), len(table), _ptr(idx), float(z_met), _ptr(x),
                                     _stream(dev)))
    return x


def stereo_tied_rows(raw_all, ml, mr):
    """filter_outputs' mask (reference process.py:319-327) as the list of kept pair rows, computed on the device:
    raw_all (ml*mr, C) -> int32 device tensor of row numbers (several per tied left person, none for a NaN one)."""
    dev = _require_cuda(raw_all.device)
    raw_all = _dev_f32(raw_all, dev)
    rows = torch.empty((ml * mr,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_stereo_tied_rows(_ptr(raw_all), int(raw_all.shape[1]), int(ml), int(mr), _ptr(rows), _ptr(count),
                                              _stream(dev)))
    return rows[:int(count.item())]


def val_stats(raw, labels):
    """The reference Trainer's host statistics (trainer.py:163-165, 213-232) of raw outputs (m, 9|10) against labels, in one
    launch on the device -> dict (see ml_val_stats in include/monoloco_hip.h)."""
    dev = _require_cuda(raw.device)
    raw = _dev_f32(raw, dev)
    labels = _dev_f32(labels, dev)
    assert raw.shape[0] == labels.shape[0]
    vals = (ctypes.c_double * 14)()
    with torch.cuda.device(dev):
        check(_lib.load().ml_val_stats(_ptr(raw), int(raw.shape[1]), _ptr(labels), int(labels.shape[1]), int(raw.shape[0]), vals,
                                       _stream(dev)), train=True)
    names = ('d', 'x', 'y', 'h', 'w', 'l', 'ori', 'aux', 'd_val', 'ori_rad', 'bi', 'bi%', 'std', 'aux_acc')
    out = {k: vals[i] for i, k in enumerate(names)}
    out['ori_val'] = out.pop('ori_rad') * 180 / 3.14        # the reference's 3.14 (losses.py:93)
    return out


def gather_rows(src, idx):
    """src[idx] for a 2-D device tensor and an int64 index tensor on the same device (ml_gather_rows)."""
    dev = _require_cuda(src.device)
    assert src.dim() == 2 and src.dtype == torch.float32 and src.is_contiguous()
    idx = idx.to(device=dev, dtype=torch.int64).contiguous()
    out = torch.empty((idx.shape[0], src.shape[1]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(_lib.load().ml_gather_rows(_ptr(src), int(src.shape[1]), _ptr(idx), int(idx.shape[0]), _ptr(out), _stream(dev)),
              train=True)
    return out


def stereo_pairs(xl, xr):
    lib = _lib.load()
    dev = _require_cuda(xl.device)
    xl = _dev_f32(xl, dev)
    xr = _dev_f32(xr, dev)
    rows = torch.empty((xl.shape[0] * xr.shape[0], 68), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.ml_stereo_pairs(_ptr(xl), xl.shape[0], _ptr(xr), xr.shape[0], _ptr(rows), _stream(dev)))
    return rows


def extract_outputs_device(raw, centre=None, kk=None, box_conf=None, row_index=None):
    """raw (m, 9|10) device tensor -> packed (m,16) and parity tensor (m,5) on the device."""
    lib = _lib.load()
    dev = _require_cuda(raw.device)
    raw = _dev_f32(raw, dev)
    m = raw.shape[0] if row_index is None else row_index.shape[0]
    out = torch.empty((m, _lib.ML_OUT_STRIDE), dtype=torch.float32, device=dev)
    xyzds = torch.empty((m, _lib.ML_XYZDS_STRIDE), dtype=torch.float32, device=dev)
    kinv = inverse_intrinsics(kk) if kk is not None else None
    centre = _dev_f32(centre, dev) if centre is not None else None
    box_conf = _dev_f32(box_conf, dev) if box_conf is not None else None
    if row_index is not None:
        row_index = row_index.to(device=dev, dtype=torch.int32).contiguous()
    with torch.cuda.device(dev):
        check(lib.ml_extract_outputs(_ptr(raw), raw.shape[1], _ptr(row_index), m, _ptr(centre),
                                     fptr(kinv) if kinv is not None else None, _ptr(box_conf), _ptr(out),
                                     _ptr(xyzds), _stream(dev)))
    return out, xyzds


def post_geometry(kps, kk, d=None, device=None, out=None):
    """(m,3,17) keypoints, K, predicted distances (m) -> (m,12) device tensor: uv_shoulder, uv_head, uv_center,
    xy_center (3), xyz_pred (3) -- the geometry of Loco.post_process in one launch (ml_post_geometry).  `d` may be a
    strided 1-D device view (a column of the packed result); `out`: a preallocated contiguous (m,12) device tensor."""
    lib = _lib.load()
    dev = _require_cuda(device)
    kps = _dev_f32(kps, dev)
    assert kps.dim() == 3 and kps.shape[1] == 3 and kps.shape[2] == 17, "keypoints must be (m, 3, 17)"
    m = kps.shape[0]
    stride = 1
    if d is not None:
        if isinstance(d, torch.Tensor) and d.dim() == 1 and d.is_cuda and d.dtype == torch.float32 and d.device == dev \
                and d.shape[0] == m and m > 0 and d.stride(0) >= 1:
            stride = int(d.stride(0))
        else:
            d = _dev_f32(d, dev).reshape(-1)
        assert d.shape[0] == m
    if out is None:
        out = torch.empty((m, 12), dtype=torch.float32, device=dev)
    assert out.is_contiguous() and tuple(out.shape) == (m, 12) and out.device == dev
    with torch.cuda.device(dev):
        check(lib.ml_post_geometry_strided(_ptr(kps), m, fptr(inverse_intrinsics(kk)), _ptr(d), stride, _ptr(out), _stream(dev)))
    return out


def extract_outputs_mono_device(raw):
    """raw (m, 9) legacy 'monoloco_p' outputs -> packed (m,16) device tensor (ml_extract_outputs_mono)."""
    lib = _lib.load()
    dev = _require_cuda(raw.device)
    raw = _dev_f32(raw, dev)
    assert raw.dim() == 2 and raw.shape[1] == 9, "extract_outputs_mono needs (m, 9) outputs"
    out = torch.empty((raw.shape[0], _lib.ML_OUT_STRIDE), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.ml_extract_outputs_mono(_ptr(raw), raw.shape[0], _ptr(out), _stream(dev)))
    return out


def laplace_sampling_device(mu_b, n_samples, seed=1):
    """(m,2) = (mu, b) -> (n_samples, m) draws of Laplace(mu, |b|) on the device (ml_laplace_sampling)."""
    lib = _lib.load()
    dev = _require_cuda(mu_b.device)
    mu_b = _dev_f32(mu_b, dev)
    assert mu_b.dim() == 2 and mu_b.shape[1] == 2
    out = torch.empty((int(n_samples), mu_b.shape[0]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        check(lib.ml_laplace_sampling(_ptr(mu_b), mu_b.shape[0], int(n_samples), int(seed), _ptr(out), _stream(dev)))
    return out


def debug_linear(x, w, b, relu=False, res=None, precision='f16x2', small_path=False, tile_kernel=None, ksplit=1, mid_dma=True):
    """Single dense layer through the MFMA kernel (test hook): the tile kernel (tile_kernel: None = the default choice, 'pp' =
    dense_kernel_pp, 'w4' = dense_kernel_w4 wherever it runs, 'mid64' / 'mid128' = dense_mid_kernel with that tile height, 'half' =
    dense_kernel_w4's half-size 256 x 128 tile for K > 128), or
    the small-row kernels.  ksplit 2 | 4 (with 'mid64' / 'mid128'): the reduction of every output tile cut into that many k ranges,
    one workgroup each, the last arriver runs the epilogue (dense_mid_kernel<.., SPLITK>)."""
    lib = _lib.load()
    dev = _require_cuda(x.device)
    x = _dev_f32(x, dev)
    w = np.ascontiguousarray(np.asarray(w, dtype=np.float32))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float32))
    n, k = w.shape
    y = torch.empty((x.shape[0], n), dtype=torch.float32, device=dev)
    res = _dev_f32(res, dev) if res is not None else None
    with torch.cuda.device(dev):
        check(lib.ml_debug_linear(_ptr(x), x.shape[0], k, fptr(w), fptr(b), n, int(bool(relu)), _ptr(res), _ptr(y),
                                  PRECISIONS[precision] | (_lib.ML_DEBUG_SMALL_PATH if small_path else 0)
                                  | {None: 0, 'pp': _lib.ML_DEBUG_TILE_PP, 'w4': _lib.ML_DEBUG_TILE_W4, 'mid64': _lib.ML_DEBUG_MID_64,
                                     'mid128': _lib.ML_DEBUG_MID_128, 'half': _lib.ML_DEBUG_MID_64 | _lib.ML_DEBUG_MID_128}[tile_kernel]
                                  | {1: 0, 2: _lib.ML_DEBUG_MID_SPLIT2, 4: _lib.ML_DEBUG_MID_SPLIT4}[int(ksplit)]
                                  | (0 if mid_dma else _lib.ML_DEBUG_MID_NODMA),
                                  _stream(dev)))
    return y


# ------------------------------------------------------------------ the model engine
class LocoEngine:
    """One packed LocoModel on one HIP device (handle of ``ml_loco_*``).

    ``state_dict`` uses the reference's parameter names (reference
    monoloco/network/architectures.py:24-43); tensors may live anywhere, they are read once.
    """

    def __init__(self, state_dict, device=None, precision='f16x2', merge_w2w3=True, reserve_rows=0):
        self._h = None
        lib = _lib.load()
        self.device = _require_cuda(device)
        if precision not in PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(PRECISIONS))
        self.precision = precision
        self.merge_w2w3 = bool(merge_w2w3)
        sd = {k: v for k, v in state_dict.items()}
        w1 = sd['w1.weight']
        self.hidden, self.in_features = int(w1.shape[0]), int(w1.shape[1])
        # LocoModel: w_fin (out-1 rows) + w_aux; legacy MonolocoModel (architectures.py:105-145): w2 maps to the outputs
        self.legacy = 'w_fin.weight' not in sd and 'w3.weight' not in sd
        self.out_features = int(sd['w2.weight'].shape[0]) if self.legacy else int(sd['w_fin.weight'].shape[0]) + 1
        self.num_stage = len({k.split('.')[1] for k in sd if k.startswith('linear_stages.')})
        handle = ctypes.c_void_p()
        check(lib.ml_loco_create(self.in_features, self.hidden, self.out_features, self.num_stage,
                                 ctypes.byref(handle)))
        self._h = handle
        for key, val in sd.items():
            if key.endswith('num_batches_tracked'):
                continue
            arr = np.ascontiguousarray(val.detach().to('cpu', torch.float32).numpy() if isinstance(val, torch.Tensor)
                                       else np.asarray(val, dtype=np.float32))
            check(lib.ml_loco_set_tensor(self._h, key.encode(), fptr(arr), arr.size))
        with torch.cuda.device(self.device):
            check(lib.ml_loco_finalize(self._h, PRECISIONS[precision],
                                       _lib.ML_FLAG_MERGE_W2W3 if self.merge_w2w3 else 0))
            if reserve_rows:
                check(lib.ml_loco_reserve(self._h, int(reserve_rows)))

    # -- lifetime
    def close(self):
        if self._h is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize(self.device)
            _lib.load().ml_loco_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def set_tuning(self, small_rows=-1, small32_rows=-1, chunk_rows=-1, tile_kernel=-1, everywhere=False, mid_rows=-1, mid_tile=-1):
        """Path selection of THIS engine (ml_loco_set_tuning; negative = unchanged): rows <= small_rows run the small-row dense
        kernels, above small32_rows with 32x32 tiles; chunk_rows > 0 walks the batch in row chunks; tile_kernel 4 (default) =
        dense_kernel_w4 for the long-K layers + dense_kernel_pp for the input / fused-head layers, 2 = dense_kernel_pp
        everywhere; everywhere=True with 4 = dense_kernel_w4 for every layer it supports; small_rows < rows <= mid_rows run
        dense_mid_kernel (mid_tile 0 = by row count, 64, 128) or, mid_tile 256 / rows > 4096, dense_kernel_w4's half-size tile."""
        tk = int(tile_kernel) | (256 if everywhere else 0) if tile_kernel >= 0 else -1
        check(_lib.load().ml_loco_set_tuning(self._h, int(small_rows), int(small32_rows), int(chunk_rows), tk, int(mid_rows),
                                             int(mid_tile)))

    def reserve(self, rows):
        with torch.cuda.device(self.device):
            check(_lib.load().ml_loco_reserve(self._h, int(rows)))

    @property
    def device_bytes(self):
        return int(_lib.load().ml_loco_device_bytes(self._h))

    # -- hot calls (asynchronous on the current stream of self.device)
    def forward_raw(self, x, out=None):
        dev = self.device
        x = _dev_f32(x, dev)
        assert x.dim() == 2 and x.shape[1] == self.in_features
        m = x.shape[0]
        raw = out if out is not None else torch.empty((m, self.out_features), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.load().ml_loco_forward_raw(self._h, _ptr(x), m, _ptr(raw), _stream(dev)))
        return raw

    def forward_mono(self, kps, kinv, box_conf=None, want_raw=False, out=None, xyzds=None, raw=None):
        """kps (m,3,17) device fp32; kinv = inverse_intrinsics(K).  Returns (out (m,16), xyzds (m,5), raw|None)."""
        dev = self.device
        kps = _dev_f32(kps, dev)
        m = kps.shape[0]
        if out is None:
            out = torch.empty((m, _lib.ML_OUT_STRIDE), dtype=torch.float32, device=dev)
        if xyzds is None:
            xyzds = torch.empty((m, _lib.ML_XYZDS_STRIDE), dtype=torch.float32, device=dev)
        if want_raw and raw is None:
            raw = torch.empty((m, self.out_features), dtype=torch.float32, device=dev)
        box_conf = _dev_f32(box_conf, dev) if box_conf is not None else None
        with torch.cuda.device(dev):
            check(_lib.load().ml_loco_forward_mono(self._h, _ptr(kps), m, fptr(kinv), _ptr(box_conf), _ptr(raw),
                                                   _ptr(out), _ptr(xyzds), _stream(dev)))
        return out, xyzds, raw

    def forward_stereo(self, kps_l, kps_r, kinv, box_conf=None, want_raw_all=False, out=None):
        """All-vs-all stereo.  Returns dict(out, xyzds, best, ties, raw_all)."""
        dev = self.device
        kps_l = _dev_f32(kps_l, dev)
        kps_r = _dev_f32(kps_r, dev)
        ml, mr = kps_l.shape[0], kps_r.shape[0]
        if out is None:
            out = torch.empty((ml, _lib.ML_OUT_STRIDE), dtype=torch.float32, device=dev)
        xyzds = torch.empty((ml, _lib.ML_XYZDS_STRIDE), dtype=torch.float32, device=dev)
        best = torch.empty((ml,), dtype=torch.int32, device=dev)
        ties = torch.zeros((1,), dtype=torch.int32, device=dev)
        raw_all = torch.empty((ml * mr, self.out_features), dtype=torch.float32, device=dev) if want_raw_all else None
        box_conf = _dev_f32(box_conf, dev) if box_conf is not None else None
        with torch.cuda.device(dev):
            check(_lib.load().ml_loco_forward_stereo(self._h, _ptr(kps_l), ml, _ptr(kps_r), mr, fptr(kinv),
                                                     _ptr(box_conf), _ptr(raw_all), _ptr(out), _ptr(xyzds),
                                                     _ptr(best), _ptr(ties), _stream(dev)))
        return dict(out=out, xyzds=xyzds, best=best, ties=ties, raw_all=raw_all)

    def epistemic_mono(self, kps, kinv, n_dropout, p_dropout=0.2, n_samples=100, seed=1, want_passes=False):
        """MC-dropout spread of the distance (reference net.py:135-161) -> (m,) device tensor
        [, (n_dropout, m, out) raw outputs of every stochastic pass]."""
        dev = self.device
        kps = _dev_f32(kps, dev)
        m = kps.shape[0]
        epi = torch.empty((m,), dtype=torch.float32, device=dev)
        passes = torch.empty((n_dropout, m, self.out_features), dtype=torch.float32, device=dev) if want_passes else None
        with torch.cuda.device(dev):
            check(_lib.load().ml_loco_epistemic_mono(self._h, _ptr(kps), m, fptr(kinv), int(n_dropout), float(p_dropout),
                                                     int(n_samples), int(seed), _ptr(epi), _ptr(passes), _stream(dev)))
        return (epi, passes) if want_passes else epi

    def epistemic_inputs(self, inputs, n_dropout, p_dropout=0.2, n_samples=100, seed=1):
        """The same spread from PRE-PROCESSED (m, 34) network inputs: the argument of the reference's
        Loco.epistemic_uncertainty(inputs) (net.py:135-161) -> (m,) device tensor."""
        dev = self.device
        x = _dev_f32(inputs, dev)
        assert x.dim() == 2 and x.shape[1] == self.in_features, "inputs must be (m, %d)" % self.in_features
        m = x.shape[0]
        epi = torch.empty((m,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.load().ml_loco_epistemic_inputs(self._h, _ptr(x), m, int(n_dropout), float(p_dropout), int(n_samples),
                                                       int(seed), _ptr(epi), None, _stream(dev)))
        return epi

    ROUTES = ('small16', 'small32', 'mid64', 'mid128', 'half', 'tile')

    def route_for_rows(self, rows):
        """Name of the dense kernel family a forward of `rows` network rows takes on this engine (ml_loco_route)."""
        code = int(_lib.load().ml_loco_route(self._h, int(rows)))
        return self.ROUTES[code] if 0 <= code < len(self.ROUTES) else 'unknown'

    def set_option(self, name, value):
        """Named switches of the route plan (ml_loco_set_option): 'half_heads', 'half_from'."""
        check(_lib.load().ml_loco_set_option(self._h, name.encode(), int(value)))

    def plan_for_rows(self, rows, mc_dropout=False, with_post=True):
        """The launch plan of a forward of `rows` network rows (ml_loco_plan): 'route=tile; L0 pp; L1 w4; ...; L6 w4+aux;
        L7 pp+fin8; end=tail_mono' -- per dense layer its kernel family and which head rides in its epilogue, then how the
        call ends.  The library executes exactly this plan."""
        buf = ctypes.create_string_buffer(1024)
        check(_lib.load().ml_loco_plan(self._h, int(rows), int(bool(mc_dropout)), int(bool(with_post)), buf, 1024))
        return buf.value.decode()

    # -- measurement
    def profile_begin(self, max_launches=65536):
        check(_lib.load().ml_loco_profile_begin(self._h, int(max_launches)))

    def profile_end(self):
        """-> dict(launches, total_ms, per_layer_ms, per_layer_n) of the dense-kernel launches since begin."""
        nl = self.num_layers
        n = ctypes.c_int64()
        tot = ctypes.c_double()
        pl = (ctypes.c_double * nl)()
        pn = (ctypes.c_int64 * nl)()
        check(_lib.load().ml_loco_profile_end(self._h, ctypes.byref(n), ctypes.byref(tot), pl, pn, nl))
        return dict(launches=int(n.value), total_ms=float(tot.value), per_layer_ms=list(pl), per_layer_n=list(pn))

    # -- test hooks
    def folded_layer(self, idx):
        lib = _lib.load()
        n, k, e = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(lib.ml_debug_get_layer(self._h, idx, None, None, ctypes.byref(n), ctypes.byref(k), ctypes.byref(e)))
        w = np.empty((n.value, k.value), dtype=np.float32)
        b = np.empty((n.value,), dtype=np.float32)
        check(lib.ml_debug_get_layer(self._h, idx, fptr(w), fptr(b), None, None, None))
        return w, b, e.value

    @property
    def num_layers(self):
        return int(_lib.load().ml_debug_num_layers(self._h))
