"""ctypes front-end of the ``ml_trainer_*`` C ABI: one LocoModel in train mode on one HIP device."""
import ctypes

import numpy as np
import torch

from .. import _lib
from .._lib import check, fptr
from ..engine import _dev_f32, _ptr, _require_cuda, _stream

TASKS = ('d', 'x', 'y', 'h', 'w', 'l', 'ori', 'aux')


ROUTES = {'auto': 0, 'exact': 1, 'mid': 2, 'fast': 3}
ROUTE_NAMES = {-1: None, 0: 'exact', 1: 'fast', 2: 'mid'}


class HipTrainer:
    """Parameters, Adam state and the training step live in the library; tensors cross by state_dict key."""

    def __init__(self, state_dict, p_dropout=0.2, lr=0.002, sched_gamma=0.98, sched_step=30, seed=1, device=None,
                 auto_tune_mtl=False, lambdas=None, route='auto', fast_rows=None, dw_layout=None):
        self._h = None
        lib = _lib.load()
        self.device = _require_cuda(device)
        w1 = state_dict['w1.weight']
        self.hidden, self.in_features = int(w1.shape[0]), int(w1.shape[1])
        self.out_features = int(state_dict['w_fin.weight'].shape[0]) + 1
        self.num_stage = len({k.split('.')[1] for k in state_dict if k.startswith('linear_stages.')})
        self.shapes = {k: tuple(v.shape) for k, v in state_dict.items() if not k.endswith('num_batches_tracked')}
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.ml_trainer_create(self.in_features, self.hidden, self.out_features, self.num_stage, float(p_dropout),
                                        float(lr), float(sched_gamma), int(sched_step), int(seed), ctypes.byref(h)), train=True)
            self._h = h
            self.load_state_dict(state_dict)
        self.set_route(route, fast_rows)
        if dw_layout is None:   # MONOLOCO_TRAIN_DW_LAYOUT=0|1|2|3 picks it without touching the caller (the reference's Trainer has no such argument)
            import os
            dw_layout = os.environ.get('MONOLOCO_TRAIN_DW_LAYOUT')
        if dw_layout is not None:
            self.set_dw_layout(int(dw_layout))
        self.last_plain = None
        self.auto_tune_mtl = bool(auto_tune_mtl)
        if self.auto_tune_mtl:   # AutoTuneMultiTaskLoss (reference train/losses.py:17-43)
            check(lib.ml_trainer_set_auto_tune(self._h, 1), train=True)
        if lambdas is not None:  # task weights (reference Trainer.lambdas, trainer.py:42), order d, x, y, h, w, l, ori[, aux]
            vals = [float(v) for v in lambdas] + [1.0] * 8
            assert not self.auto_tune_mtl or all(v in (0.0, 1.0) for v in vals[:8]), "auto-tune needs lambdas in {0, 1} (losses.py:21)"
            check(lib.ml_trainer_set_lambdas(self._h, (ctypes.c_float * 8)(*vals[:8])), train=True)

    def set_route(self, route='auto', fast_rows=None):
        """Which GEMM route the steps of THIS trainer take (ml_trainer_set_route): 'auto' (>= fast_rows rows, default 4096:
        the large-batch 3-product kernels; below: the mid route of csrc/train_mid.h; else exact fp32), 'exact', 'mid', 'fast'."""
        check(_lib.load().ml_trainer_set_route(self._h, ROUTES[route], -1 if fast_rows is None else int(fast_rows)), train=True)

    def set_dw_layout(self, dw_layout):
        """Large-batch route (>= fast_rows rows) only.  1 (the default): the residual stream, the stages' inner activations and dz exist
        between kernels as fp16 hi+lo lines only (~22 significant bits, re-rounded once per stage; dz scaled by a bound of its column
        maxima) -- 1.5 ms faster per 65536-row step; the parity numbers of tests/test_gpu_train.py are for this layout.  0 / 2: those
        tensors stay fp32 between kernels like the reference's (0 = transposed operand copies, 2 = reduction-major operands; same bits).
        Round 6: layout 1 also runs w2 -> w3 as ONE Linear (three batch-sized GEMMs instead of six, include/monoloco_hip.h); 3 = layout 1
        with the two Linears apart (the A/B reference)."""
        assert int(dw_layout) in (0, 1, 2, 3)
        check(_lib.load().ml_trainer_set_tuning(self._h, 0, -1, int(dw_layout)), train=True)

    @property
    def last_route(self):
        """'exact' | 'fast' | 'mid': the route the last step took."""
        return ROUTE_NAMES[int(_lib.load().ml_trainer_last_route(self._h))]

    def debug_read(self, which, shape):
        """Bring-up: an internal fp32 buffer (ml_trainer_debug_read) as a CPU tensor of `shape`."""
        arr = np.empty(shape, dtype=np.float32)
        with torch.cuda.device(self.device):
            check(_lib.load().ml_trainer_debug_read(self._h, int(which), fptr(arr), arr.size), train=True)
        return torch.from_numpy(arr)

    @property
    def log_sigmas(self):
        """The learnable log_sigma per task (d, x, y, h, w, l, ori[, aux]) of the auto-tuned loss; zeros when it is off."""
        buf = (ctypes.c_float * 8)()
        check(_lib.load().ml_trainer_get_log_sigmas(self._h, buf), train=True)
        return torch.tensor(list(buf)[:self.out_features - 2])

    def set_log_sigmas(self, values):
        vals = [float(v) for v in values] + [0.0] * 8
        buf = (ctypes.c_float * 8)(*vals[:8])
        check(_lib.load().ml_trainer_set_log_sigmas(self._h, buf), train=True)

    def close(self):
        if self._h is not None:
            _lib.load().ml_trainer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state_dict(self, state_dict):
        self.version = getattr(self, 'version', 0) + 1   # weights / statistics changed: whoever caches derived state keys on this
        lib = _lib.load()
        with torch.cuda.device(self.device):
            for key, val in state_dict.items():
                if key.endswith('num_batches_tracked'):
                    continue
                if isinstance(val, torch.Tensor) and val.is_cuda and val.device == self.device:
                    # already on the trainer's device: device to device (ml_trainer_set_tensor copies with hipMemcpyDefault)
                    dv = val.detach().to(torch.float32).contiguous()
                    check(lib.ml_trainer_set_tensor(self._h, key.encode(), ctypes.cast(dv.data_ptr(), ctypes.POINTER(ctypes.c_float)),
                                                    dv.numel()), train=True)
                    continue
                arr = np.ascontiguousarray(val.detach().to('cpu', torch.float32).numpy() if isinstance(val, torch.Tensor)
                                           else np.asarray(val, dtype=np.float32))
                check(lib.ml_trainer_set_tensor(self._h, key.encode(), fptr(arr), arr.size), train=True)

    def _get(self, fn, key):
        arr = np.empty(self.shapes[key], dtype=np.float32)
        with torch.cuda.device(self.device):
            check(fn(self._h, key.encode(), fptr(arr), arr.size), train=True)
        return torch.from_numpy(arr)

    # ---- flat exchange (ml_trainer_copy_flat): all parameters / gradients / running statistics in ONE device-to-device copy each
    def _flat_layout(self):
        if getattr(self, '_layout', None) is None:
            lib = _lib.load()
            n_p, n_s = ctypes.c_int64(), ctypes.c_int64()
            check(lib.ml_trainer_flat_numel(self._h, ctypes.byref(n_p), ctypes.byref(n_s)), train=True)
            lay = {}
            for k, shape in self.shapes.items():
                isp = ctypes.c_int()
                off = int(lib.ml_trainer_flat_offset(self._h, k.encode(), ctypes.byref(isp)))
                assert off >= 0, k
                lay[k] = (off, int(np.prod(shape)) if len(shape) else 1, bool(isp.value))
            self._layout = (lay, int(n_p.value), int(n_s.value))
        return self._layout

    def _copy_flat(self, what, flat):
        with torch.cuda.device(self.device):
            check(_lib.load().ml_trainer_copy_flat(self._h, int(what), _ptr(flat), flat.numel(), _stream(self.device)), train=True)

    def load_tensors_flat(self, tensors):
        """`tensors`: {state_dict key: CUDA tensor on this device} covering EVERY parameter and running statistic: two flat
        device-to-device copies instead of one synchronising copy per tensor."""
        lay, n_p, n_s = self._flat_layout()
        flat_p = torch.empty((n_p,), dtype=torch.float32, device=self.device)
        flat_s = torch.empty((n_s,), dtype=torch.float32, device=self.device)
        for k, (off, n, isp) in lay.items():
            (flat_p if isp else flat_s)[off:off + n].copy_(tensors[k].detach().reshape(-1))
        self._copy_flat(0, flat_p)
        self._copy_flat(3, flat_s)
        self.version += 1

    def grads_flat(self):
        """{key: device tensor (a view of one flat buffer)} of every parameter gradient."""
        lay, n_p, _ = self._flat_layout()
        flat = torch.empty((n_p,), dtype=torch.float32, device=self.device)
        self._copy_flat(2, flat)
        return {k: flat[off:off + n].view(self.shapes[k]) for k, (off, n, isp) in lay.items() if isp}

    def stats_flat(self):
        """{key: device tensor} of the BatchNorm running statistics."""
        lay, _, n_s = self._flat_layout()
        flat = torch.empty((n_s,), dtype=torch.float32, device=self.device)
        self._copy_flat(4, flat)
        return {k: flat[off:off + n].view(self.shapes[k]) for k, (off, n, isp) in lay.items() if not isp}

    def _get_device(self, fn, key):
        """The same tensor as a DEVICE tensor on the trainer's device (no host round trip)."""
        out = torch.empty(self.shapes[key], dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(fn(self._h, key.encode(), ctypes.cast(out.data_ptr(), ctypes.POINTER(ctypes.c_float)), out.numel()), train=True)
        return out

    def grads_device(self):
        """grads() as device tensors."""
        lib = _lib.load()
        return {k: self._get_device(lib.ml_trainer_get_grad, k) for k in self.shapes if 'running_' not in k}

    def state_dict(self):
        lib = _lib.load()
        return {k: self._get(lib.ml_trainer_get_tensor, k) for k in self.shapes}

    def grads(self):
        lib = _lib.load()
        return {k: self._get(lib.ml_trainer_get_grad, k) for k in self.shapes if 'running_' not in k}

    @property
    def num_steps(self):
        return int(_lib.load().ml_trainer_num_steps(self._h))

    def step(self, inputs, labels, update=True, want_outputs=False):
        """One training step (reference trainer.py:154-161).  Returns dict(loss, d, x, y, h, w, l, ori, aux)
        [and the train-mode outputs]."""
        dev = self.device
        x = _dev_f32(inputs, dev)
        y = _dev_f32(labels, dev)
        m = x.shape[0]
        assert y.shape[0] == m and x.shape[1] == self.in_features
        losses = (ctypes.c_double * 9)()
        raw = torch.empty((m, self.out_features), dtype=torch.float32, device=dev) if want_outputs else None
        with torch.cuda.device(dev):
            check(_lib.load().ml_trainer_step(self._h, _ptr(x), _ptr(y), int(y.shape[1]), m, int(bool(update)), losses,
                                              _ptr(raw), _stream(dev)), train=True)
            self.version += 1   # (a step without update still moves the BatchNorm running statistics)
            vals = (ctypes.c_double * 10)()
            check(_lib.load().ml_trainer_last_val_values(self._h, vals), train=True)
        out = {'loss': losses[0]}
        out.update({t: losses[1 + i] for i, t in enumerate(TASKS)})
        self.last_plain = self._plain(vals)
        return (out, raw) if want_outputs else out

    def forward_train(self, inputs):
        """The train-mode forward on its own (reference trainer.py:155 `outputs = self.model(inputs)`; ml_trainer_forward_train):
        batch-statistics BatchNorm (running statistics updated), fresh dropout masks -> (m, out_features) device tensor.  `inputs`
        must stay alive and unchanged until backward() has run (the returned tensor keeps a reference)."""
        dev = self.device
        x = _dev_f32(inputs, dev)
        assert x.dim() == 2 and x.shape[1] == self.in_features
        raw = torch.empty((x.shape[0], self.out_features), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(_lib.load().ml_trainer_forward_train(self._h, _ptr(x), int(x.shape[0]), _ptr(raw), _stream(dev)), train=True)
        self.version += 1          # (the running statistics moved)
        self._pending_x = x
        self.forward_seq = getattr(self, 'forward_seq', 0) + 1   # (which forward the workspace holds: _TrainForward checks it)
        return raw

    def backward(self, grad_outputs):
        """`loss.backward()` for the outputs of the last forward_train (trainer.py:158; ml_trainer_backward): grad_outputs (m,
        out_features) = gradient of the caller's loss with respect to them.  The UNCLIPPED parameter gradients are then in grads()."""
        dev = self.device
        g = _dev_f32(grad_outputs, dev)
        assert g.dim() == 2 and g.shape[1] == self.out_features
        with torch.cuda.device(dev):
            check(_lib.load().ml_trainer_backward(self._h, _ptr(g), int(g.shape[0]), _stream(dev)), train=True)
        self._pending_x = None

    @staticmethod
    def _plain(vals):
        """The unweighted means of one batch: training-type task values d (Laplace), x, y, h, w, l, ori, aux and the two
        validation-type ones that differ (reference losses.py:85-96): 'd_val' = L1 on d, 'ori_val' = angle error in degrees."""
        out = {t: vals[i] for i, t in enumerate(TASKS)}
        out['d_val'] = vals[8]
        out['ori_val'] = vals[9] * 180 / 3.14
        return out

    def evaluate_batch(self, inputs, labels, want_outputs=False):
        """Eval-mode forward (running statistics, no dropout) of the trainer's current weights on one batch, on the device
        (ml_trainer_eval): dict of the training-type task means d (Laplace), x, y, h, w, l, ori, aux and the validation-type
        'd_val' (L1), 'ori_val' (degrees) [and the raw outputs].  Needs hidden % 64 == 0 (MonolocoHipError otherwise)."""
        dev = self.device
        x = _dev_f32(inputs, dev)
        y = _dev_f32(labels, dev)
        m = x.shape[0]
        vals = (ctypes.c_double * 10)()
        raw = torch.empty((m, self.out_features), dtype=torch.float32, device=dev) if want_outputs else None
        with torch.cuda.device(dev):
            check(_lib.load().ml_trainer_eval(self._h, _ptr(x), _ptr(y), int(y.shape[1]), m, vals, _ptr(raw), _stream(dev)), train=True)
        out = self._plain(vals)
        return (out, raw) if want_outputs else out

    @property
    def can_evaluate(self):
        """evaluate_batch runs for this shape (the mid route's kernels: hidden % 64 == 0)."""
        return bool(_lib.load().ml_trainer_can_eval(self._h))   # the library's own predicate (mid_possible, csrc/train.hip)

    def snapshot(self):
        """Keep the current parameters + running statistics aside on the device (the loop's best-epoch copy)."""
        with torch.cuda.device(self.device):
            check(_lib.load().ml_trainer_snapshot(self._h, _stream(self.device)), train=True)

    def restore(self):
        self.version += 1
        with torch.cuda.device(self.device):
            check(_lib.load().ml_trainer_restore(self._h, _stream(self.device)), train=True)
