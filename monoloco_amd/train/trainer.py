"""``Trainer`` with the reference's surface (monoloco/train/trainer.py:36-248) on the HIP training step.

Same constructor contract (an argparse-like ``args`` with joints, epochs, lr, bs, hidden_size, n_stage, dropout,
sched_step, sched_gamma, r_seed, mode, out, no_save), same loop (per epoch: a training pass with Adam + per-batch
StepLR + clip 3, then a validation pass; the weights of the best validation 'd' epoch are kept, trainer.py:173-177)
and ``evaluate`` saves a reference-compatible ``state_dict`` with ``torch.save``.  The dataset is the reference's
``joints-*.json`` (KeypointsDataset, train/datasets.py:49-66); shuffling uses torch's DataLoader exactly as the
reference does, so with ``--dropout 0`` the trajectory is comparable batch by batch.
"""
import copy
import datetime
import json
import math
import os
import time
from collections import defaultdict

import torch
from torch.utils.data import DataLoader, Dataset

from .. import engine
from ..network.architectures import LocoModel
from .hip_trainer import HipTrainer


class KeypointsDataset(Dataset):
    """Inputs X (N, 34|68) and labels Y (N, 10|11) of one phase of a joints json (reference datasets.py:44-96)."""

    def __init__(self, joints, phase):
        assert phase in ['train', 'val', 'test']
        with open(joints, 'r') as f:
            dic_jo = json.load(f)
        self.inputs_all = torch.tensor(dic_jo[phase]['X'])
        self.outputs_all = torch.tensor(dic_jo[phase]['Y'])
        self.names_all = dic_jo[phase]['names']
        self.kps_all = torch.tensor(dic_jo[phase]['kps'])
        self.version = dic_jo['version']
        self.dic_clst = dic_jo[phase]['clst']

    def __len__(self):
        return self.inputs_all.shape[0]

    def __getitem__(self, idx):
        return self.inputs_all[idx, :], self.outputs_all[idx], self.names_all[idx], self.kps_all[idx, :]

    def get_cluster_annotations(self, clst):
        inputs = torch.tensor(self.dic_clst[clst]['X'])
        outputs = torch.tensor(self.dic_clst[clst]['Y']).float()
        return inputs, outputs, len(self.dic_clst[clst]['Y'])

    def get_version(self):
        return self.version


class _EpochBatches:
    """The row-number batches of ``DataLoader(dataset_of_n_rows, batch_size=bs, shuffle=True)`` without the loader: per epoch
    the two draws from torch's global generator that a DataLoader makes (the iterator's base seed, then RandomSampler's seed),
    ``randperm(n)`` from a generator seeded with the second, cut into batches.  ~15 us per epoch instead of ~100 (iterator,
    sampler, per-sample fetch, collate).  It restates torch internals, so it is only used after ``matches_dataloader`` has seen it
    produce a real DataLoader's batches AND leave the global generator in the same state (Trainer falls back to the loader
    otherwise)."""

    def __init__(self, n, bs):
        self.n, self.bs = int(n), int(bs)

    def __iter__(self):
        torch.empty((), dtype=torch.int64).random_()                          # _BaseDataLoaderIter: base seed (unused here)
        seed = int(torch.empty((), dtype=torch.int64).random_().item())       # RandomSampler.__iter__
        g = torch.Generator()
        g.manual_seed(seed)
        perm = torch.randperm(self.n, generator=g)
        for lo in range(0, self.n, self.bs):
            yield perm[lo:lo + self.bs]

    def matches_dataloader(self, epochs=2):
        state = torch.get_rng_state()
        try:
            mine = [[b.tolist() for b in self] for _ in range(epochs)]
            after_mine = torch.get_rng_state()
            torch.set_rng_state(state)
            loader = DataLoader(_IndexDataset(self.n), batch_size=self.bs, shuffle=True)
            ref = [[b.tolist() for b in loader] for _ in range(epochs)]
            return mine == ref and torch.equal(after_mine, torch.get_rng_state())
        except Exception:  # any change of torch's internals: use the loader
            return False
        finally:
            torch.set_rng_state(state)


class _IndexDataset(Dataset):
    """Row numbers only: a DataLoader over it draws exactly the random numbers the reference's DataLoader over its
    KeypointsDataset draws (same sampler, same batching), so the batches are the reference's -- while the rows themselves stay
    on the device and are gathered there."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        return idx


class Trainer:
    VAL_BS = 10000
    tasks = ('d', 'x', 'y', 'h', 'w', 'l', 'ori', 'aux')
    val_task = 'd'
    lambdas = (1, 1, 1, 1, 1, 1, 1, 1)
    clusters = ['10', '20', '30', '40']
    input_size = dict(mono=34, stereo=68)
    output_size = dict(mono=9, stereo=10)

    def __init__(self, args):
        assert os.path.exists(args.joints), "Input file not found"
        self.mode = args.mode
        self.joints = args.joints
        self.num_epochs = args.epochs
        self.no_save = getattr(args, 'no_save', False)
        self.lr = args.lr
        if getattr(args, 'out', None):
            self.path_out = args.out
        else:
            name = 'monoloco_pp' if self.mode == 'mono' else 'monstereo'
            self.path_out = os.path.join('data', 'outputs', name + '-' + datetime.datetime.now().strftime("%Y%m%d-%H%M")[2:] + '.pkl')
        self.path_model = self.path_out
        self.device = torch.device('cuda')
        torch.manual_seed(args.r_seed)
        if self.mode == 'mono':
            self.tasks = self.tasks[:-1]
        datasets = {phase: KeypointsDataset(self.joints, phase=phase) for phase in ['train', 'val']}
        self.dataset_sizes = {phase: len(datasets[phase]) for phase in ['train', 'val']}
        # the reference's loaders (trainer.py:104-106: batch_size=bs, shuffle=True for BOTH phases) over row numbers; the rows
        # live on the device for the whole training
        self.dataloaders = {}
        for phase in ['train', 'val']:
            fast = _EpochBatches(self.dataset_sizes[phase], args.bs)
            self.dataloaders[phase] = fast if fast.matches_dataloader() else \
                DataLoader(_IndexDataset(self.dataset_sizes[phase]), batch_size=args.bs, shuffle=True)
        self._rows = {phase: (datasets[phase].inputs_all.to(self.device, torch.float32).contiguous(),
                              datasets[phase].outputs_all.to(self.device, torch.float32).contiguous()) for phase in ['train', 'val']}
        # same construction order as the reference (trainer.py:115-123), hence the same default initialisation
        self.model = LocoModel(input_size=self.input_size[self.mode], output_size=self.output_size[self.mode],
                               linear_size=args.hidden_size, p_dropout=args.dropout, num_stage=args.n_stage)
        # AutoTuneMultiTaskLoss instead of MultiTaskLoss (reference trainer.py:95-98)
        self.auto_tune_mtl = bool(getattr(args, 'auto_tune_mtl', False))
        self.hip = HipTrainer(self.model.state_dict(), p_dropout=args.dropout, lr=args.lr, sched_gamma=args.sched_gamma,
                              sched_step=int(args.sched_step), seed=args.r_seed, device=self.device,
                              auto_tune_mtl=self.auto_tune_mtl, lambdas=self.lambdas[:len(self.tasks)],
                              dw_layout=getattr(args, 'dw_layout', None))   # (None: library default / MONOLOCO_TRAIN_DW_LAYOUT)
        self.epoch_losses = defaultdict(lambda: defaultdict(list))
        self._eval_eng, self._eval_version = None, -1

    # ---- validation: eval-mode forward (running statistics) on the inference engine, built ONCE per weight version
    def _eval_engine(self):
        """LocoEngine of the trainer's current weights (only for shapes the trainer's own evaluation does not take: hidden % 64
        != 0); keyed on HipTrainer.version, which every step / load_state_dict / restore bumps.  w2/w3 are kept as two layers
        (merge_w2w3=False): no host fp64 H x H x H product per rebuild."""
        from ..engine import LocoEngine
        version = self.hip.version
        if self._eval_eng is None or self._eval_version != version:
            self._close_eval_engine()
            self._eval_eng = LocoEngine(self.hip.state_dict(), device=self.device, merge_w2w3=False)
            self._eval_version = version
        return self._eval_eng

    def _close_eval_engine(self):
        if getattr(self, '_eval_eng', None) is not None:
            self._eval_eng.close()
        self._eval_eng, self._eval_version = None, -1

    def _forward_eval(self, inputs):
        """Raw eval-mode outputs on the device (the engine fallback for shapes ml_trainer_eval does not take)."""
        return self._eval_engine().forward_raw(inputs.to(self.device))

    def _val_losses(self, out, lab):
        """Validation values of the reference for raw outputs `out` and labels `lab` (device tensors): per task the
        `losses_val` entries of CompositeLoss (losses.py:85-96: L1 from Laplace for d, angle error for ori, BCE for
        aux, L1 otherwise) and 'all' = the training-type multi-task loss on these outputs (losses.py:59-73 with unit
        lambdas; trainer.py:195) -- one launch on the device (ml_val_stats), for shapes the trainer's own evaluation does
        not take."""
        plain = engine.val_stats(out, lab)
        if 'aux' not in self.tasks:
            plain['aux'] = 0.0
        return self._vals(plain)

    def _vals(self, plain):
        """The reference's validation-type record of one batch from its unweighted means (HipTrainer.last_plain /
        evaluate_batch: computed by the loss kernel on the device): per task the validation value, and 'all' = the training-type
        multi-task loss."""
        vals = {'d': plain['d_val'], 'ori': plain['ori_val']}
        for t in ('x', 'y', 'h', 'w', 'l'):
            vals[t] = plain[t]
        train_type = [plain[t] for t in ('d', 'x', 'y', 'h', 'w', 'l', 'ori')]
        if 'aux' in self.tasks:
            vals['aux'] = plain['aux']
            train_type.append(plain['aux'])
        lam = self.lambdas[:len(train_type)]
        if self.auto_tune_mtl:   # losses.py:34-39: every task * lambda / (2 sigma^2), plus the log_sigmas
            ls = self.hip.log_sigmas.tolist()
            vals['all'] = sum(la * v / (2.0 * math.exp(s) ** 2) + s for la, v, s in zip(lam, train_type, ls))
        else:
            vals['all'] = sum(la * v for la, v in zip(lam, train_type))
        return vals

    def _batch(self, phase, idx):
        x_all, y_all = self._rows[phase]
        idx = idx.to(self.device)
        return engine.gather_rows(x_all, idx), engine.gather_rows(y_all, idx)

    def _eval_batch(self, x, y, want_outputs=False):
        """Validation values (and raw outputs) of the CURRENT weights on one device batch: on the trainer's own kernels where
        the shape allows (no engine rebuild, nothing leaves the device but ten numbers), else through the inference engine."""
        if self.hip.can_evaluate:
            if want_outputs:
                plain, raw = self.hip.evaluate_batch(x, y, want_outputs=True)
                return self._vals(plain), raw
            return self._vals(self.hip.evaluate_batch(x, y)), None
        out = self._forward_eval(x)
        return self._val_losses(out, y), out

    def train(self):
        since = time.time()
        self.hip.snapshot()
        best_acc, best_epoch = 1e6, 0
        self.step_seconds = 0.0          # time inside the device calls (each synchronises): bench.py's device-busy fraction
        for epoch in range(self.num_epochs):
            running = defaultdict(lambda: defaultdict(float))
            for idx in self.dataloaders['train']:
                inputs, labels = self._batch('train', idx)
                # the reference logs, for the training phase as well, the validation-type values of the train-mode outputs it
                # has just back-propagated (trainer.py:163-165, epoch_logs :193-197): the loss kernel reduces them on the way;
                # 'loss' (the optimised total of the step) is kept beside them
                t0 = time.perf_counter()
                losses = self.hip.step(inputs, labels, update=True)
                self.step_seconds += time.perf_counter() - t0
                running['train']['loss'] += losses['loss'] * inputs.size(0)
                for k, v in self._vals(self.hip.last_plain).items():
                    running['train'][k] += v * inputs.size(0)
            for idx in self.dataloaders['val']:
                inputs, labels = self._batch('val', idx)
                t0 = time.perf_counter()
                vals, _ = self._eval_batch(inputs, labels)
                self.step_seconds += time.perf_counter() - t0
                for k, v in vals.items():
                    running['val'][k] += v * inputs.size(0)
            for phase in running:
                for k, v in running[phase].items():
                    self.epoch_losses[phase][k].append(v / self.dataset_sizes[phase])
            val_d = self.epoch_losses['val'][self.val_task][-1]   # KeyError-free: every epoch appends every task
            if val_d < best_acc:
                best_acc, best_epoch = val_d, epoch
                self.hip.snapshot()       # device-to-device: the weights of the best epoch never travel
        self.training_time = time.time() - since
        self.hip.restore()
        self._close_eval_engine()   # the weights changed without a step: never reuse the engine of the last epoch
        return best_epoch

    def evaluate(self, load=False, model=None, debug=False):
        """Reference trainer.py:197-246: statistics on the whole validation set ('all') and per distance cluster
        ('10'..'40'), the model saved as a reference-compatible state_dict and returned in eval mode."""
        if load:
            self.hip.load_state_dict(torch.load(model, map_location=lambda storage, loc: storage))
            self._close_eval_engine()
        sd = self.hip.state_dict()
        self.model.load_state_dict(sd, strict=False)
        self.model.eval()
        dataset = KeypointsDataset(self.joints, phase='val')
        dic_err = {'val': defaultdict(lambda: defaultdict(float))}
        # the sigmas = exp(log_sigma) of the auto-tuned loss, zeros otherwise (reference trainer.py:208, 281-284, printed at
        # :297-300; its index arithmetic there only works for stereo -- for mono it runs past the list -- so the values are
        # taken from the loss itself here)
        dic_err['val']['sigmas'] = [math.exp(v) for v in self.hip.log_sigmas.tolist()] if self.auto_tune_mtl else [0.] * len(self.tasks)

        def stats(inputs, labels, clst):
            labels = labels.to(self.device, torch.float32)
            vals, out = self._eval_batch(inputs.to(self.device, torch.float32), labels, want_outputs=True)
            entry = dic_err['val'][clst]
            for t in self.tasks:
                if t != 'aux':
                    entry[t] = vals[t]
            entry['all'] = vals['all']
            # bi = exp(s) d (unnormalize_bi, process.py:125-133), the share of |mu - d| <= bi, torch's unbiased std of the errors and
            # the aux accuracy (get_accuracy, trainer.py:384-389): reduced on the device from the rows that are there anyway
            ext = engine.val_stats(out, labels)
            entry['bi'] = ext['bi']
            entry['bi%'] = ext['bi%']
            entry['std'] = torch.tensor(ext['std'], dtype=torch.float32)         # the reference stores the 0-dim tensor of .std()
            entry['aux'] = 0 if self.mode == 'mono' else ext['aux_acc']

        inputs, labels, _, _ = dataset[0:len(dataset)]
        stats(inputs, labels, 'all')
        for clst in self.clusters:
            if clst in dataset.dic_clst and len(dataset.dic_clst[clst]['Y']):
                c_in, c_lab, _ = dataset.get_cluster_annotations(clst)
                stats(c_in, c_lab, clst)
        self._close_eval_engine()
        if not (self.no_save or load):
            torch.save({k: v.clone() for k, v in self.model.state_dict().items()}, self.path_model)
        return dic_err, self.model
