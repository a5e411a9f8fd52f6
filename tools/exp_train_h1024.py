"""Per-tensor deviation of the first training step at hidden 1024 from the reference-loop golden (tests/golden/golden_train_h1024.npz),
every route that accepts the batch: max error / max|ref|, rms error / rms(ref), elements beyond 3e-4 of max|ref|, and the
reference's own fp32-vs-fp64 deviation.  Separates broad rounding noise from the few elements a flipped ReLU mask moves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
G = os.path.join(ROOT, 'tests', 'golden')
g = dict(np.load(os.path.join(G, 'golden_train_h1024.npz')))
inp = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
dev = torch.device('cuda', 0)
for tag, routes in (('r512', ('exact', 'mid')), ('r4096', ('exact', 'mid', 'fast'))):
    m, seed = [int(v) for v in g[tag + '_rows_seed']]
    xb, yb = synth.big_train_batch(inp['mono_x'], inp['mono_y'], m, seed)
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(seed, 34, 9, 1024).items()}
    res = {}
    for route in routes:
        tr = HipTrainer(sd0, p_dropout=0.0, lr=0.001, device=dev, route=route)
        tr.step(torch.tensor(xb), torch.tensor(yb), update=False)
        res[route] = {k: v.numpy() for k, v in tr.grads().items()}
        tr.close()
    print('==', tag, ' per route: rms error / rms(ref) against the reference fp32 run | against its fp64 run ;  last columns: the reference\'s own fp32-vs-fp64 rms and max')
    for k in res[routes[0]]:
        ref = g[tag + '_grad0/' + k]
        ref64 = g[tag + '_grad0_f64/' + k].astype(np.float64)
        gmax = float(g[tag + '_gmax/' + k])
        if gmax < 1e-12:
            continue
        line = '%-38s' % k
        for route in routes:
            mine = res[route][k]
            if mine.shape != ref.shape:
                mine = mine[::64]
            r32 = np.sqrt(np.mean((mine.astype(np.float64) - ref) ** 2)) / max(np.sqrt(np.mean(ref.astype(np.float64) ** 2)), 1e-30)
            r64 = np.sqrt(np.mean((mine.astype(np.float64) - ref64) ** 2)) / max(np.sqrt(np.mean(ref64 ** 2)), 1e-30)
            line += ' | %s %8.2e %8.2e' % (route[0], r32, r64)
        print(line + ' | ref %8.2e %8.2e' % (float(g[tag + '_noise_rms/' + k]), float(g[tag + '_noise/' + k])))
