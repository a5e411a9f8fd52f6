"""Bring-up of the training step's mid route (csrc/train_mid.h): buffer-by-buffer comparison with the exact-fp32 route on
the same batch (forward activations through ml_trainer_debug_read, outputs, loss gradient, parameter gradients), then ms per
step of every route at the reference's batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))


def compare(hidden, m, p_drop, mode='mono'):
    in_f, out_f = (34, 9) if mode == 'mono' else (68, 10)
    x, y = g[mode + '_x'], g[mode + '_y']
    if m != len(x):
        x, y = synth.big_train_batch(x, y, m, 3)
    x, y = torch.tensor(x), torch.tensor(y)
    S = 3
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(33, in_f, out_f, hidden).items()}
    names = ['a%d' % s for s in range(S + 1)] + ['t%d' % s for s in range(S)] + ['z0'] + [n for s in range(S) for n in ('za%d' % s, 'zb%d' % s)] + ['z3', 'y2', 'y3']
    bufs, res = {}, {}
    for route in ('exact', 'mid'):
        tr = HipTrainer(sd, p_dropout=p_drop, lr=0.001, device=dev, seed=3, route=route)
        r, out = tr.step(x, y, update=False, want_outputs=True)
        bufs[route] = {n: tr.debug_read(i, (m, hidden)) for i, n in enumerate(names)}
        bufs[route]['out'] = out.cpu()
        bufs[route]['dout'] = tr.debug_read(201, (m, out_f))
        res[route] = (r, tr.grads(), tr.last_route)
        tr.close()
    print('hidden %d rows %d p_drop %.1f %s  routes %s %s  loss %.6f %.6f' % (hidden, m, p_drop, mode, res['exact'][2], res['mid'][2],
                                                                           res['exact'][0]['loss'], res['mid'][0]['loss']))
    for n in names + ['out', 'dout']:
        a, b = bufs['exact'][n], bufs['mid'][n]
        print('   %-5s max|exact| %.3e  max|diff| %.3e  nan %d' % (n, a.abs().max().item(), (a - b).abs().max().item(), int(torch.isnan(b).sum())))
    d = (bufs['exact']['za0'] - bufs['mid']['za0']).abs()
    print('   za0 max|diff| per 128-row block:', ' '.join('%.0e' % d[i:i + 128].max().item() for i in range(0, m, 128)))
    print('   za0 max|diff| per 128-column block:', ' '.join('%.0e' % d[:, j:j + 128].max().item() for j in range(0, hidden, 128)))
    for k in res['exact'][1]:
        a, b = res['exact'][1][k], res['mid'][1][k]
        print('   grad %-40s max|exact| %.3e  rel diff %.3e' % (k, a.abs().max().item(), ((a - b).abs().max() / a.abs().max().clamp_min(1e-30)).item()))


def timing():
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
    for m in (64, 128, 256, 331, 512, 1024, 2048, 4096):
        xb, yb = synth.big_train_batch(g['mono_x'], g['mono_y'], m, 3)
        x, y = torch.tensor(xb).to(dev), torch.tensor(yb).to(dev)
        out = {}
        for name in ('exact', 'mid', 'mid1s', 'mid2l', 'fast'):
            tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev, route=name[:3] if name.startswith('mid') else name)
            from monoloco_amd import _lib
            if name == 'mid1s':
                _lib.check(_lib.load().ml_trainer_set_tuning(tr._h, 0, 1, -1), train=True)
            if name == 'mid2l':
                _lib.check(_lib.load().ml_trainer_set_tuning(tr._h, 0, 2, -1), train=True)
            for _ in range(5): tr.step(x, y)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 30
            for _ in range(n): tr.step(x, y)
            torch.cuda.synchronize(); out[name] = (time.perf_counter() - t0) / n * 1e3
            tr.close()
        print('rows %5d: exact %.3f ms  mid %.3f ms (two launches per Linear %.3f, with the side stream %.3f)  fast %.3f ms' % (
            m, out['exact'], out['mid'], out['mid2l'], out['mid1s'], out['fast']), flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('all', 'compare'):
        compare(128, 331, 0.0)
        compare(1024, 331, 0.2)
        compare(256, 700, 0.0, 'stereo')
    if what == 'rows':   # where (if anywhere) the mid route drifts from the exact route as the batch grows
        for m in [int(a) for a in sys.argv[2:]] or [1024, 2048, 4096]:
            compare(1024, m, 0.0)
    if what in ('all', 'timing'):
        timing()
