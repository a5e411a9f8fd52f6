"""-m gpu: LocoModel as the reference uses it in a training loop -- `outputs = self.model(inputs)` in train mode, the caller's own
criterion, `loss.backward()`, `clip_grad_norm_`, a torch optimizer (monoloco/train/trainer.py:150-161, train/hyp_tuning.py) -- on the
HIP training kernels through ml_trainer_forward_train / ml_trainer_backward (torch.autograd.Function in
monoloco_amd/network/architectures.py).  Checked against the oracle's torch-CPU restatement of the same module (oracle/train_oracle.py)."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _batch(mode):
    g = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
    return torch.tensor(g[mode + '_x']), torch.tensor(g[mode + '_y'])


def _module(sd, in_f, out_f, hidden, p_dropout, dev):
    from monoloco_amd.network.architectures import LocoModel
    m = LocoModel(in_f, out_f, hidden, p_dropout=p_dropout)
    m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.parametrize("mode,in_f,out_f,hidden,rows", [('mono', 34, 9, 256, None), ('stereo', 68, 10, 128, None), ('mono', 34, 9, 1024, 5000)])
def test_train_mode_forward_and_backward_match_the_oracle(hip_lib, cuda_device, mode, in_f, out_f, hidden, rows):
    """One call of the module in train mode (dropout 0) and loss.backward() with the reference's MultiTaskLoss restated in torch:
    outputs, every parameter's .grad and the BatchNorm running statistics against the oracle's autograd run (fp32 and fp64).
    331-row fixture batch (exact-fp32 route) and a 5000-row batch at hidden 1024 (the 3-product large-batch route)."""
    from oracle.train_oracle import forward_train, multitask_loss
    x, y = _batch(mode)
    if rows:
        xb, yb = synth.big_train_batch(x.numpy(), y.numpy(), rows, 9)
        x, y = torch.tensor(xb), torch.tensor(yb)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(17, in_f, out_f, hidden).items()}
    model = _module(sd, in_f, out_f, hidden, 0.0, cuda_device).train()
    xd, yd = x.to(cuda_device), y.to(cuda_device)
    out = model(xd)
    assert out.requires_grad and out.shape == (x.shape[0], out_f) and out.device.type == 'cuda'
    loss, _ = multitask_loss(out, yd)
    loss.backward()
    # the oracle: the same functional forward under torch autograd, fp64 (exact) and fp32 (the reference's arithmetic)
    res = {}
    for dt in (torch.float64, torch.float32):
        params = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items() if 'running_' not in k and not k.endswith('num_batches_tracked')}
        run = {k: v.to(dt).clone() for k, v in sd.items() if 'running_' in k}
        o = forward_train(params, run, x.to(dt), 0.0, 3)
        l, _ = multitask_loss(o, y.to(dt))
        l.backward()
        res[dt] = (o.detach(), float(l.detach()), {k: v.grad for k, v in params.items()}, run)
    o64, l64, g64, run64 = res[torch.float64]
    o32, l32, g32, _ = res[torch.float32]
    assert abs(float(loss.detach()) - l64) <= 2e-5 * max(1.0, abs(l64))
    noise = float((o32.double() - o64).abs().max())
    assert float((out.detach().cpu().double() - o64).abs().max()) <= 2 * noise + 2e-5
    gmax = max(float(v.abs().max()) for v in g64.values())
    m_rows = x.shape[0]
    for k, p_ in model.named_parameters():
        assert p_.grad is not None and p_.grad.device == p_.device, k
        ref = g64[k]
        if float(ref.abs().max()) <= 1e-9 * gmax:       # a Linear bias in front of a BatchNorm: mathematically zero
            assert float(p_.grad.abs().max()) <= 2e-7 * gmax, k
            continue
        scale = float(ref.abs().max())
        e = float((p_.grad.cpu().double() - ref).abs().max()) / scale
        e32 = float((g32[k].double() - ref).abs().max()) / scale
        assert e <= max(8 * e32, 3e-3, 5.0 / m_rows), (k, e, e32)
    for name, buf in model.named_buffers():             # momentum-0.1 running statistics of the batch, as torch's BatchNorm1d updates them
        if name.endswith(('running_mean', 'running_var')):
            assert float((buf.cpu().double() - run64[name]).abs().max()) <= 1e-4 * max(1.0, float(run64[name].abs().max())), name
        elif name.endswith('num_batches_tracked'):
            assert int(buf) == 1, name
    # eval mode afterwards is the engine again, on the updated statistics
    model.eval()
    assert not model(xd[:16]).requires_grad


def test_a_caller_owned_training_loop_learns(hip_lib, cuda_device):
    """The reference's loop body with torch's own Adam, StepLR and clip_grad_norm_ on the module's nn.Parameters (trainer.py:127-131,
    150-161), dropout 0.2 from the library's generator: the loss falls over 30 iterations of the fixture batch, different dropout masks
    are drawn per call, zero_grad / accumulation behave like torch's."""
    from oracle.train_oracle import multitask_loss
    x, y = _batch('mono')
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(18, 34, 9, 256).items()}
    model = _module(sd, 34, 9, 256, 0.2, cuda_device).train()
    xd, yd = x.to(cuda_device), y.to(cuda_device)
    with torch.no_grad():
        a, b = model(xd), model(xd)
    assert not torch.equal(a, b)                        # fresh masks per call
    opt = torch.optim.Adam(model.parameters(), lr=0.001)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=30, gamma=0.98)
    first = last = None
    for it in range(30):
        opt.zero_grad()
        loss, _ = multitask_loss(model(xd), yd)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
        opt.step()
        sched.step()
        first = float(loss) if first is None else first
        last = float(loss)
    assert np.isfinite(last) and last < 0.7 * first, (first, last)
    # gradients accumulate across two backward calls like any autograd graph's
    opt.zero_grad()
    model.dropout.p = 0.2
    l1, _ = multitask_loss(model(xd), yd)
    l1.backward()
    g1 = model.w_fin.weight.grad.clone()
    l2, _ = multitask_loss(model(xd), yd)
    l2.backward()
    assert float((model.w_fin.weight.grad - g1).abs().max()) > 0 and torch.isfinite(model.w_fin.weight.grad).all()
    # the input's gradient is not provided: said so, not silently None
    xr = xd.clone().requires_grad_(True)
    with pytest.raises(NotImplementedError, match="inputs"):
        model(xr)


def test_a_backward_behind_a_second_forward_is_refused(hip_lib, cuda_device):
    """The library keeps ONE train-mode forward's activations: a backward whose forward has been overwritten by another train-mode
    forward of the module must say so (torch's autograd would differentiate the right batch; differentiating the wrong one silently is
    the failure to exclude), and the later forward's own backward still works."""
    from oracle.train_oracle import multitask_loss
    x, y = _batch('mono')
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(19, 34, 9, 256).items()}
    model = _module(sd, 34, 9, 256, 0.0, cuda_device).train()
    xd, yd = x.to(cuda_device), y.to(cuda_device)
    out1 = model(xd)
    out2 = model(xd * 1.01)
    with pytest.raises(RuntimeError, match="another train-mode forward"):
        multitask_loss(out1, yd)[0].backward()
    multitask_loss(out2, yd)[0].backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    # the plain C entry points refuse a backward without its forward, and one behind an evaluation through the same workspace
    tr = model._hip_trainer(cuda_device)
    out3 = tr.forward_train(xd)
    tr.evaluate_batch(xd, yd)
    with pytest.raises(Exception, match="pending"):
        tr.backward(torch.ones_like(out3))
