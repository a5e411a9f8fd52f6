"""CPU oracle of the training step (TEST INFRASTRUCTURE): a torch-autograd restatement of one iteration of the
reference's loop, monoloco/train/trainer.py:150-161 -- LocoModel forward in train mode
(network/architectures.py:48-102), MultiTaskLoss over the tasks d, x, y, h, w, l, ori[, aux]
(train/losses.py:59-73, 80-83, 112-131; label/ output slices network/process.py:240-249, 293-301),
clip_grad_norm_(3), Adam(lr) and a StepLR stepped per batch (trainer.py:128-131).  Pinned against the real
reference by tests/golden/golden_train.npz (oracle/make_golden.py)."""
import torch
import torch.nn.functional as F


def _bn_train(x, sd, name, run, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, run[name + '.running_mean'], run[name + '.running_var'], sd[name + '.weight'],
                        sd[name + '.bias'], True, momentum, eps)


def forward_train(params, run, x, p_dropout=0.0, num_stage=3):
    lin = lambda t, n: F.linear(t, params[n + '.weight'], params[n + '.bias'])
    drop = lambda t: F.dropout(t, p_dropout, True)
    y = drop(torch.relu(_bn_train(lin(x, 'w1'), params, 'batch_norm1', run)))
    for i in range(num_stage):
        p = 'linear_stages.%d.' % i
        t = drop(torch.relu(_bn_train(lin(y, p + 'w1'), params, p + 'batch_norm1', run)))
        t = drop(torch.relu(_bn_train(lin(t, p + 'w2'), params, p + 'batch_norm2', run)))
        y = y + t
    y = lin(y, 'w2')
    aux = lin(y, 'w_aux')
    y = drop(torch.relu(_bn_train(lin(y, 'w3'), params, 'batch_norm3', run)))
    return torch.cat((lin(y, 'w_fin'), aux), dim=1)


def multitask_loss(out, lab):
    """Sum of the task losses, all lambdas 1; returns (total, dict)."""
    mu, si, xx = out[:, 2:3], out[:, 3:4], lab[:, 3:4]
    norm = 1 - mu / xx
    vals = {'d': torch.mean(torch.abs(norm) * torch.exp(-si) + 0.01 + si + 2)}
    for t, c in (('x', 0), ('y', 1), ('h', 4), ('w', 5), ('l', 6)):
        vals[t] = F.l1_loss(out[:, c:c + 1], lab[:, c:c + 1])
    vals['ori'] = F.l1_loss(out[:, 7:9], lab[:, 7:9])
    if out.shape[1] == 10:
        vals['aux'] = F.binary_cross_entropy_with_logits(out[:, 9:10], lab[:, 10:11])
    return sum(vals.values()), vals


class OracleTrainer:
    def __init__(self, state_dict, lr, p_dropout=0.0, sched_step=30, sched_gamma=0.98, dtype=torch.float32, auto_tune_mtl=False,
                 lambdas=None):
        self.params = {k: torch.as_tensor(v).to(dtype).clone().requires_grad_(True) for k, v in state_dict.items()
                       if 'running_' not in k and not k.endswith('num_batches_tracked')}
        self.run = {k: torch.as_tensor(v).to(dtype).clone() for k, v in state_dict.items() if 'running_' in k}
        self.p = p_dropout
        self.num_stage = len({k.split('.')[1] for k in state_dict if k.startswith('linear_stages.')})
        # AutoTuneMultiTaskLoss (reference train/losses.py:17-43): one learnable log_sigma per task, in the same optimiser
        # (trainer.py:127-129 chains mt_loss.parameters()), outside clip_grad_norm_ (trainer.py:159: model.parameters())
        n_tasks = 8 if self.params['w_fin.weight'].shape[0] == 9 else 7
        self.lambdas = [float(v) for v in (lambdas if lambdas is not None else [1.0] * 8)][:n_tasks]   # trainer.py:42
        self.log_sigmas = torch.zeros(n_tasks, dtype=dtype, requires_grad=True) if auto_tune_mtl else None
        self.opt = torch.optim.Adam(list(self.params.values()) + ([self.log_sigmas] if auto_tune_mtl else []), lr=lr)
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=sched_step, gamma=sched_gamma)

    def step(self, x, lab, update=True):
        self.opt.zero_grad()
        out = forward_train(self.params, self.run, x, self.p, self.num_stage)
        loss, vals = multitask_loss(out, lab)
        vals = {k: lam * v for (k, v), lam in zip(vals.items(), self.lambdas)}      # losses.py:66 / :34
        loss = sum(vals.values())
        if self.log_sigmas is not None:
            vals = {k: v / (2.0 * (s.exp() ** 2)) for (k, v), s in zip(vals.items(), self.log_sigmas)}
            loss = sum(vals.values()) + self.log_sigmas.sum()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(list(self.params.values()), 3)
        if update:
            self.opt.step()
            self.sched.step()
        res = {'loss': float(loss.detach())}
        res.update({k: float(v.detach()) for k, v in vals.items()})
        return res, out.detach()

    def grads(self):
        return {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v)) for k, v in self.params.items()}

    def state_dict(self):
        sd = {k: v.detach().clone() for k, v in self.params.items()}
        sd.update({k: v.clone() for k, v in self.run.items()})
        return sd
