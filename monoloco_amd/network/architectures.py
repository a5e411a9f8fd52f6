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
            raise NotImplementedError(
                "%s.forward in train mode: this module's forward is the eval-mode network on the HIP engine (running-stat "
                "BatchNorm, no dropout, no autograd graph) -- the reference's `outputs = self.model(inputs); loss.backward()` "
                "(monoloco/train/trainer.py:155-161) has no counterpart on the module.  Train with monoloco_amd.train.Trainer "
                "(same arguments and checkpoints as monoloco.train.Trainer; monoloco_amd.train.HipTrainer is the step underneath), "
                "or call monoloco_amd.compat.install(trainer=True) so that `monoloco.train.Trainer` IS that class; call .eval() "
                "for inference; MC-dropout runs through Loco(n_dropout=...)" % type(self).__name__)
        home = x.device
        eng = self.hip_engine(home if home.type == 'cuda' else None)
        return eng.forward_raw(x.detach()).to(home)


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
