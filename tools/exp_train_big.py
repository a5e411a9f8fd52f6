"""Per-tensor deviation of the HIP training step (exact / mid / fast route, HipTrainer(route=...)) from the reference's big-batch golden."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
G = os.path.join(ROOT, 'tests', 'golden')
g = dict(np.load(os.path.join(G, 'golden_train_big.npz')))
inp = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
dev = torch.device('cuda', 0)
for mode, in_f, out_f in (('mono', 34, 9), ('stereo', 68, 10)):
    m, seed = [int(v) for v in g[mode + '_rows_seed']]
    xb, yb = synth.big_train_batch(inp[mode + '_x'], inp[mode + '_y'], m, seed)
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(seed, in_f, out_f, 256).items()}
    for name in ('exact', 'mid', 'fast'):
        tr = HipTrainer(sd0, p_dropout=0.0, lr=0.001, device=dev, route=name)
        res, out = tr.step(torch.tensor(xb), torch.tensor(yb), want_outputs=True, update=False)
        print(mode, name, 'out err %.2e (max %.1f)' % (np.abs(out.cpu().numpy() - g[mode + '_out0']).max(), np.abs(g[mode + '_out0']).max()),
              'loss', res['loss'], g[mode + '_loss0'][0])
        for k, v in tr.grads().items():
            key = mode + '_grad0/' + k
            if key in g:
                ref = g[key]
                print('   %-40s max|ref| %.2e  err %.2e  rel %.2e' % (k, np.abs(ref).max(), np.abs(v.numpy() - ref).max(),
                                                                      np.abs(v.numpy() - ref).max() / np.abs(ref).max()))
        tr.close()
