"""``LocoModel`` / ``MonolocoModel`` parameter containers with the reference's state_dict layout
(reference monoloco/network/architectures.py), whose eval forward runs on the HIP engine.

The modules own ordinary ``nn.Linear`` / ``nn.BatchNorm1d`` parameters under the reference's names
(``w1``, ``batch_norm1``, ``linear_stages.{i}.w1`` ...), so reference checkpoints load with
``load_state_dict`` unchanged.  ``forward`` in eval mode packs the current parameters once
(BN folded, fp16 hi|lo split, uploaded) and calls the MFMA kernels; there is no torch compute path.
The training step (batch-stat BN, dropout, backward, Adam) lives in ``monoloco_amd.train`` (ml_trainer_*).
"""
import torch
from torch import nn

from .. import engine


class _Stage(nn.Module):
    """One residual stage: y = x + relu(bn2(w2(relu(bn1(w1 x))))) (reference architectures.py:74-102)."""

    def __init__(self, size, p_dropout):
        super().__init__()
        self.l_size = size
        self.w1 = nn.Linear(size, size)
        self.batch_norm1 = nn.BatchNorm1d(size)
        self.w2 = nn.Linear(size, size)
        self.batch_norm2 = nn.BatchNorm1d(size)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)


class _HipForward(nn.Module):
    """Eval-mode forward on the HIP engine, shared by LocoModel and the legacy MonolocoModel."""
    precision = 'f16x2'
    merge_w2w3 = True
    _engine = None
    _engine_key = None

    def __getstate__(self):
        # copy.deepcopy(model) / torch.save(model): the native handles (the packed engine, the train-mode twin) are per-object caches that
        # hold raw pointers -- a copy starts without them and rebuilds them on first use
        state = self.__dict__.copy()
        for k in ('_engine', '_engine_key', '_hip_tr', '_hip_trainer_key'):
            state.pop(k, None)
        return state

    # -- engine management: re-pack when parameters were replaced or modified in place
    def _params_key(self, dev):
        return (str(dev), self.precision, self.merge_w2w3) + tuple(
            (t.data_ptr(), t._version) for t in self.state_dict().values())

    def hip_engine(self, device=None):
        dev = engine._require_cuda(device)
        key = self._params_key(dev)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = engine.LocoEngine(self.state_dict(), device=dev, precision=self.precision,
                                             merge_w2w3=self.merge_w2w3)
            self._engine_key = key
        return self._engine

    def forward(self, x):
        if self.training or self.dropout.training:
            return self._forward_train(x)
        home = x.device
        eng = self.hip_engine(home if home.type == 'cuda' else None)
        return eng.forward_raw(x.detach()).to(home)

    def _forward_train(self, x):
        raise NotImplementedError(
            "%s.forward in train mode: this module's forward is the eval-mode network on the HIP engine (running-stat "
            "BatchNorm, no dropout, no autograd graph); only LocoModel has a train-mode forward.  Train with monoloco_amd.train.Trainer "
            "(same arguments and checkpoints as monoloco.train.Trainer; monoloco_amd.train.HipTrainer is the step underneath), "
            "or call monoloco_amd.compat.install(trainer=True) so that `monoloco.train.Trainer` IS that class; call .eval() "
            "for inference; MC-dropout runs through Loco(n_dropout=...)" % type(self).__name__)


class _TrainForward(torch.autograd.Function):
    """`outputs = self.model(inputs)` / `loss.backward()` of a caller-owned training loop (reference trainer.py:155-158,
    hyp_tuning.py) on the HIP training kernels: forward = ml_trainer_forward_train (LocoModel.forward in train mode,
    architectures.py:48-71), backward = ml_trainer_backward from the caller's gradient of the outputs; the parameter gradients come
    back as the Function's input gradients, so `loss.backward()`, `clip_grad_norm_` and any torch optimizer work on the module's own
    nn.Parameters.  Dropout masks come from the library's counter-based generator (statistically, not bitwise, torch's)."""

    @staticmethod
    def forward(ctx, module, x, *params):
        tr = module._hip_trainer(x.device if x.is_cuda else None)
        ctx.tr, ctx.keys, ctx.home = tr, [k for k, _ in module.named_parameters()], [p_.device for p_ in params]
        out = tr.forward_train(x.detach())
        ctx.seq = tr.forward_seq
        # BatchNorm's running statistics moved inside the library: bring them back into the module's buffers
        with torch.no_grad():
            stats = tr.stats_flat()            # (one stream-ordered device-to-device copy)
            for name, buf in module.named_buffers():
                if name.endswith(('running_mean', 'running_var')):
                    buf.copy_(stats[name].to(buf.device))
                elif name.endswith('num_batches_tracked'):
                    buf += 1
        module._hip_trainer_key = module._train_key()     # (the buffers just written are the trainer's own values)
        return out.to(x.device)

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.seq != ctx.tr.forward_seq:
            # the library keeps ONE forward's activations (the reference's autograd keeps one graph per call): a second train-mode
            # forward of the module has overwritten what this backward needs -- say so instead of differentiating the wrong batch
            raise RuntimeError("LocoModel (train mode): another train-mode forward ran before this backward; the HIP trainer keeps the "
                               "activations of the last forward only -- call backward() before the next forward, or use eval mode / "
                               "torch.no_grad() for forwards that are not differentiated")
        ctx.tr.backward(grad_out.contiguous())
        g = ctx.tr.grads_flat()        # ONE device-to-device copy: the 34 MB of gradients never visit the host
        return (None, None) + tuple(g[k].to(dev_) for k, dev_ in zip(ctx.keys, ctx.home))


def _lib_load():
    from .. import _lib
    return _lib.load()


class LocoModel(_HipForward):
    """MonoLoco++ (input 34) / MonStereo (input 68) residual MLP (reference architectures.py:6-71).

    ``output_size`` counts the auxiliary head as in the reference: w_fin has output_size-1 rows and
    w_aux one."""

    def __init__(self, input_size, output_size=2, linear_size=512, p_dropout=0.2, num_stage=3, device='cuda',
                 precision='f16x2', merge_w2w3=True):
        super().__init__()
        self.stereo_size = input_size
        self.mono_size = int(input_size / 2)
        self.output_size = output_size - 1
        self.linear_size = linear_size
        self.p_dropout = p_dropout
        self.num_stage = num_stage
        self.device = device
        self.precision = precision
        self.merge_w2w3 = merge_w2w3
        self.w1 = nn.Linear(input_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList([_Stage(linear_size, p_dropout) for _ in range(num_stage)])
        self.w2 = nn.Linear(linear_size, linear_size)
        self.w3 = nn.Linear(linear_size, linear_size)
        self.batch_norm3 = nn.BatchNorm1d(linear_size)
        self.w_aux = nn.Linear(linear_size, 1)
        self.w_fin = nn.Linear(linear_size, self.output_size)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)


    # ---- train mode: an autograd-capable forward (round 6; the reference trains by calling the module, trainer.py:155-161)
    _hip_tr = None
    _hip_trainer_key = None

    def _train_key(self):
        return tuple((t.data_ptr(), t._version) for t in self.state_dict().values())

    def _hip_trainer(self, device=None):
        """The library-side twin of this module in train mode: created once, its parameters and running statistics refreshed
        whenever the module's tensors were replaced or modified in place (an optimizer step, load_state_dict)."""
        from ..train.hip_trainer import HipTrainer
        dev = engine._require_cuda(device)
        if self._hip_tr is not None and self._hip_tr.device != dev:
            self._hip_tr.close()
            self._hip_tr = None
        if self._hip_tr is None:
            self._hip_tr = HipTrainer(self.state_dict(), p_dropout=self.p_dropout, device=dev)
            self._hip_trainer_key = self._train_key()
        elif self._hip_trainer_key != self._train_key():
            sd = self.state_dict()
            if all(v.is_cuda and v.device == dev for k, v in sd.items() if not k.endswith('num_batches_tracked')):
                self._hip_tr.load_tensors_flat(sd)        # two flat device-to-device copies (after every optimizer step)
            else:
                self._hip_tr.load_state_dict(sd)
            self._hip_trainer_key = self._train_key()
        return self._hip_tr

    def _forward_train(self, x):
        if self.training != self.dropout.training and self.p_dropout > 0:
            # BatchNorm in train mode with dropout switched off (or the reverse): the library's train-mode forward has one switch
            raise NotImplementedError("LocoModel: BatchNorm and Dropout must be in the same mode (model.train() / model.eval())")
        assert x.dim() == 2 and x.shape[1] == self.stereo_size, "inputs must be (m, %d)" % self.stereo_size
        if x.requires_grad:
            raise NotImplementedError("LocoModel in train mode: the gradient with respect to the inputs is not computed (the "
                                      "reference's inputs are data, trainer.py:152)")
        return _TrainForward.apply(self, x, *[p_ for _, p_ in self.named_parameters()])


class MonolocoModel(_HipForward):
    """Legacy MonoLoco (hidden 256, 2 outputs: d and log(b/d); reference architectures.py:105-176): the same
    w1/bn/relu + residual stages as LocoModel, followed by one Linear ``w2`` to the outputs.  Runs on the same
    dense kernels; ``w2`` is a GEMV-shaped head."""

    def __init__(self, input_size, output_size=2, linear_size=256, p_dropout=0.2, num_stage=3):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.linear_size = linear_size
        self.p_dropout = p_dropout
        self.num_stage = num_stage
        self.w1 = nn.Linear(input_size, linear_size)
        self.batch_norm1 = nn.BatchNorm1d(linear_size)
        self.linear_stages = nn.ModuleList([_Stage(linear_size, p_dropout) for _ in range(num_stage)])
        self.w2 = nn.Linear(linear_size, output_size)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
