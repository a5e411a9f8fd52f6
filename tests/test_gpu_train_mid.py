"""-m gpu: the training step's "mid" route (monoloco_amd/csrc/train_mid.h: the reference's real batch sizes, run.py:95
--bs 512) -- its GEMM on its own against fp64, the whole step against the exact-fp32 route, against the reference's own
loop at the headline width (tests/golden/golden_train_h1024.npz, oracle/make_golden.py train_h1024) and the optimizer's
W^T bookkeeping."""
import ctypes
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _batch(mode, val=False):
    g = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
    s = 'val' if val else ''
    return torch.tensor(g[mode + '_x' + s]), torch.tensor(g[mode + '_y' + s])


def _tgemm(lib, dev, a, b, tile_rows, bias=None, res=None, amax=None, bmax=None, want_ct=False):
    from monoloco_amd._lib import check
    from monoloco_amd.engine import _ptr, _stream
    M, K = a.shape
    N = b.shape[0]
    c = torch.full((M, N), float('nan'), dtype=torch.float32, device=dev)
    ldct = (M + 31) // 32 * 32
    ct = torch.full((N, ldct), float('nan'), dtype=torch.float32, device=dev) if want_ct else None
    with torch.cuda.device(dev):
        check(lib.ml_debug_tgemm(_ptr(a), _ptr(b), _ptr(c), M, N, K, _ptr(bias), _ptr(res), _ptr(amax), _ptr(bmax), _ptr(ct), ldct,
                                 tile_rows, _stream(dev)), train=True)
    torch.cuda.synchronize()
    return c, ct


@pytest.mark.parametrize("M,N,K,tile", [(331, 1024, 1024, 32), (331, 1024, 1024, 64), (1024, 1024, 352, 64), (70, 128, 96, 32),
                                        (1, 64, 32, 32), (513, 256, 1056, 64), (512, 192, 64, 32)])
def test_tgemm_against_fp64(hip_lib, cuda_device, M, N, K, tile):
    """c = a . b^T on the 3-product fp16 MFMA scheme with fp32 operands split on the fly: fp32-class accuracy (the error of
    one product is ~2^-22 of |a| |b|), edge rows, odd numbers of k-steps, both tile shapes."""
    dev = cuda_device
    gen = torch.Generator().manual_seed(M * 7 + K)
    a = (torch.randn(M, K, generator=gen) * torch.rand(M, 1, generator=gen) * 3).to(dev)
    b = (torch.randn(N, K, generator=gen) * 0.05).to(dev)
    ref = a.double() @ b.double().t()
    mag = a.double().abs() @ b.double().abs().t()
    c, _ = _tgemm(hip_lib, dev, a, b, tile)
    err = ((c.double() - ref).abs() / mag).max().item()
    assert torch.isfinite(c).all() and err <= 2.0e-6, err          # 2^-22 * a few accumulation roundings
    # an fp32 torch matmul of the same operands is not closer to fp64 than this kernel by more than a small factor
    e32 = ((a @ b.t()).double() - ref).abs().max().item()
    assert (c.double() - ref).abs().max().item() <= max(8 * e32, 1e-6 * mag.max().item())


def test_tgemm_epilogue_and_scales(hip_lib, cuda_device):
    """bias, residual (aliasing the output is allowed), operand scale words (small gradients: without the scale their fp16
    halves would be subnormal), transposed copy with zero padding."""
    dev = cuda_device
    gen = torch.Generator().manual_seed(5)
    M, N, K = 203, 128, 160
    a = (torch.randn(M, K, generator=gen) * 3e-6).to(dev)            # a gradient-sized operand
    b = (torch.randn(N, K, generator=gen) * 0.03).to(dev)
    bias = torch.randn(N, generator=gen).to(dev) * 1e-6
    res = (torch.randn(M, N, generator=gen) * 1e-6).to(dev)
    amax = a.abs().max().reshape(1).clone()
    bmax = b.abs().max().reshape(1).clone()
    ref = a.double() @ b.double().t() + bias.double() + res.double()
    mag = a.double().abs() @ b.double().abs().t()
    for tile in (32, 64):
        c, ct = _tgemm(hip_lib, dev, a, b, tile, bias=bias, res=res, amax=amax, bmax=bmax, want_ct=True)
        err = ((c.double() - ref).abs() / mag).max().item()
        assert err <= 3e-6, (tile, err)
        assert torch.equal(ct[:, :M], c.t()) and (ct[:, M:] == 0).all(), tile
        c0, _ = _tgemm(hip_lib, dev, a, b, tile, bias=bias, res=res)      # unscaled: visibly worse (subnormal lo halves)
        err0 = ((c0.double() - ref).abs() / mag).max().item()
        assert err0 > 4 * err, (err0, err)


@pytest.mark.parametrize("mode,hidden,p_drop,rows", [('mono', 256, 0.0, None), ('stereo', 128, 0.2, None), ('mono', 1024, 0.2, None),
                                                     ('mono', 1024, 0.0, 512), ('stereo', 320, 0.0, 1500)])
def test_mid_route_matches_exact_route(hip_lib, cuda_device, mode, hidden, p_drop, rows):
    """Same step on the exact-fp32 route and on the mid route: losses, outputs, gradients, BatchNorm statistics; the device
    RNG is the same on both, so dropout masks agree.  (hidden 320: % 64 == 0 but not % 256 -- mid is the only fast route.)"""
    from monoloco_amd.train import HipTrainer
    in_f, out_f = (34, 9) if mode == 'mono' else (68, 10)
    x, y = _batch(mode)
    if rows:
        xb, yb = synth.big_train_batch(x.numpy(), y.numpy(), rows, 9)
        x, y = torch.tensor(xb), torch.tensor(yb)
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(33, in_f, out_f, hidden).items()}
    got = {}
    for name in ('exact', 'mid'):
        tr = HipTrainer(sd0, p_dropout=p_drop, lr=0.001, device=cuda_device, seed=3, route=name)
        res, out = tr.step(x, y, update=False, want_outputs=True)
        assert tr.last_route == name
        sd = tr.state_dict()
        got[name] = (res, out.cpu().numpy(), {k: v.numpy() for k, v in tr.grads().items()},
                     {k: v.numpy() for k, v in sd.items() if 'running' in k})
        res2 = tr.step(x, y)          # and a real update step runs
        assert np.isfinite(res2['loss'])
        tr.close()
    (r0, o0, g0, s0), (r1, o1, g1, s1) = got['exact'], got['mid']
    assert not np.array_equal(o0, o1), "the mid route did not run"
    assert np.abs(o0 - o1).max() <= 2e-5 * max(1.0, np.abs(o0).max()), np.abs(o0 - o1).max()
    for k in r0:
        assert abs(r0[k] - r1[k]) <= 1e-4 * max(1.0, abs(r0[k])), (k, r0[k], r1[k])
    for k in s0:
        assert np.abs(s0[k] - s1[k]).max() <= 1e-5 * max(1.0, np.abs(s0[k]).max()), k
    gmax = max(np.abs(v).max() for v in g0.values())
    for k in g0:   # (ReLU masks of pre-activations within rounding of 0 flip between two fp32-class implementations)
        scale = max(np.abs(g0[k]).max(), 1e-4 * gmax)
        assert np.abs(g0[k] - g1[k]).max() / scale <= 3e-3, (k, np.abs(g0[k] - g1[k]).max() / scale)


def test_mid_route_keeps_transposed_weights(hip_lib, cuda_device):
    """The optimizer writes W, W^T and max |W| together; set_tensor marks them stale.  After updates on the mid route, after
    a load_state_dict and after a step on another route the W^T images the data-gradient GEMMs read equal the weights."""
    from monoloco_amd.train import HipTrainer
    x, y = _batch('mono')
    hidden, S = 128, 3
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(35, 34, 9, hidden).items()}
    names = ['linear_stages.%d.%s.weight' % (s, w) for s in range(S) for w in ('w1', 'w2')] + ['w2.weight', 'w3.weight']

    def check_wt(tr):
        sd = tr.state_dict()
        words = tr.debug_read(400, (64,))
        for slot, name in enumerate(names):
            wt = tr.debug_read(300 + slot, (hidden, hidden))
            assert torch.equal(wt, sd[name].t().contiguous()), name
            assert words[slot].item() == sd[name].abs().max().item(), name

    tr = HipTrainer(sd0, p_dropout=0.0, lr=0.001, device=cuda_device, route='mid')
    for _ in range(3):
        tr.step(x, y)
    check_wt(tr)
    tr.step(x, y, update=False)
    check_wt(tr)
    sd1 = {k: v * 1.5 for k, v in tr.state_dict().items()}
    tr.load_state_dict(sd1)
    a = tr.step(x, y)
    check_wt(tr)
    fresh = HipTrainer(sd1, p_dropout=0.0, lr=0.001, device=cuda_device, route='mid')
    b = fresh.step(x, y)
    assert abs(a['loss'] - b['loss']) <= 1e-6 * abs(b['loss'])      # the reloaded trainer computed with the reloaded weights
    tr.set_route('exact')
    tr.step(x, y)
    tr.set_route('mid')
    tr.step(x, y)
    check_wt(tr)
    tr.close()
    fresh.close()


def test_mid_route_trajectory_tracks_exact_route(hip_lib, cuda_device):
    """Six update steps (Adam, StepLR, clip) on both routes from the same start: losses stay together and the weights end
    within a few Adam steps of each other (sign flips of gradients within rounding of zero are +-lr each)."""
    from monoloco_amd.train import HipTrainer
    x, y = _batch('mono')
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(36, 34, 9, 256).items()}
    tr = {n: HipTrainer(sd0, p_dropout=0.0, lr=0.001, sched_gamma=0.5, sched_step=2, device=cuda_device, route=n) for n in ('exact', 'mid')}
    for step in range(6):
        l0, l1 = tr['exact'].step(x, y)['loss'], tr['mid'].step(x, y)['loss']
        assert abs(l0 - l1) <= 2e-3 * max(1.0, abs(l0)), (step, l0, l1)
    s0, s1 = tr['exact'].state_dict(), tr['mid'].state_dict()
    for k in s0:
        d = (s0[k] - s1[k]).abs()
        assert d.max().item() <= 4.5e-3, (k, d.max().item())
        if k.endswith('weight') and s0[k].dim() == 2:
            assert (d > 1e-4).float().mean().item() < 0.02, (k, (d > 1e-4).float().mean().item())
    for t in tr.values():
        t.close()


@pytest.mark.parametrize("tag,route", [('r512', 'mid'), ('r4096', 'fast')])
def test_headline_width_steps_match_reference(hip_lib, cuda_device, tag, route):
    """The reference's own loop body (trainer.py:150-161, torch CPU fp32; oracle/make_golden.py train_h1024) at hidden 1024:
    a 512-row batch (run.py:95 default --bs: the mid route) and a 4096-row batch (the large-batch route), first step:
    outputs, losses, clipped gradients.  Tolerances per tensor: a multiple of the reference's OWN fp32 rounding noise
    (its fp32 run against its fp64 run, stored with the golden), with a floor."""
    from monoloco_amd.train import HipTrainer
    g = dict(np.load(os.path.join(G, 'golden_train_h1024.npz')))
    m, seed = [int(v) for v in g[tag + '_rows_seed']]
    x, y = _batch('mono')
    xb, yb = synth.big_train_batch(x.numpy(), y.numpy(), m, seed)
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(seed, 34, 9, 1024).items()}
    tr = HipTrainer(sd0, p_dropout=0.0, lr=0.001, device=cuda_device)
    res, out = tr.step(torch.tensor(xb), torch.tensor(yb), want_outputs=True)
    assert tr.last_route == route
    ref_out, ref64 = g[tag + '_out0'], g[tag + '_out0_f64']
    noise_out = np.abs(ref_out - ref64).max()
    err_out = np.abs(out.cpu().numpy() - ref_out).max()
    assert err_out <= 2.0 * noise_out + 2e-5, (err_out, noise_out)
    assert np.abs(out.cpu().numpy() - ref64).max() <= 2.0 * noise_out + 2e-5          # ... and as close to fp64 as the reference is
    names = ['loss', 'd', 'x', 'y', 'h', 'w', 'l', 'ori']
    got = np.array([res[n] for n in names])
    assert np.abs(got - g[tag + '_loss0']).max() <= 2e-5 * max(1.0, np.abs(g[tag + '_loss0']).max()), (got, g[tag + '_loss0'])
    grads = tr.grads()
    gmax_all = max(float(g[tag + '_gmax/' + k]) for k in grads)
    worst = {}
    for k, v in grads.items():
        ref_g = g[tag + '_grad0/' + k]
        gmax = float(g[tag + '_gmax/' + k])
        mine = v.numpy()
        if mine.shape != ref_g.shape:
            mine = mine[::64]                              # the 1024 x 1024 matrices are stored as every 64th row
        rel = np.abs(mine - ref_g).max() / max(gmax, 1e-4 * gmax_all)
        noise = float(g[tag + '_noise/' + k]) if gmax > 1e-9 * gmax_all else 0.0    # (biases in front of a BatchNorm: zero gradient)
        worst[k] = (rel, noise)
        assert rel <= max(3.0 * noise, 3e-4), (k, rel, noise)
    tr.close()
