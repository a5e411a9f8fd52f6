"""Round 6: the large-batch training step with w2 -> w3 as one Linear (dw_layout 1) against the two Linears apart (dw_layout 3):
ms per step, and one dropout-0 step of each against the other (losses, outputs, every gradient)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
rng = np.random.default_rng(0)
idx = rng.integers(0, len(g['mono_x']), M)
x = (torch.tensor(g['mono_x'])[idx] + torch.tensor(rng.normal(0, 0.01, (M, 34)).astype(np.float32))).to(dev)
y = torch.tensor(g['mono_y'])[idx].to(dev)
res = {}
for lay in (3, 1):
    tr = HipTrainer(sd, p_dropout=0.0, lr=0.001, device=dev)
    tr.set_dw_layout(lay)
    out = tr.step(x, y, update=False, want_outputs=True)
    losses, outs = out if isinstance(out, tuple) else (out, None)
    res[lay] = (losses, outs.clone() if outs is not None else None, {k: v.clone() for k, v in tr.grads().items()})
    tr.close()
l3, o3, g3 = res[3]; l1, o1, g1 = res[1]
print('losses  apart %s\n        merged %s' % (l3, l1))
if o3 is not None: print('outputs max abs diff %.3e (max |out| %.3e)' % ((o3 - o1).abs().max().item(), o3.abs().max().item()))
worst = 0.0
for k in g3:
    a, b = g3[k].double(), g1[k].double()
    rel = ((a - b).norm() / (a.norm() + 1e-30)).item()
    worst = max(worst, rel)
    if rel > 1e-5 or k in ('w2.weight', 'w3.weight', 'w2.bias', 'w_aux.weight', 'w3.bias'): print('  %-40s rms-rel diff %.3e' % (k, rel))
print('worst rms-rel gradient difference %.3e' % worst)
for lay in (3, 1, 3, 1):
    tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev)
    tr.set_dw_layout(lay)
    for _ in range(5): tr.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): tr.step(x, y)
    e1.record(); torch.cuda.synchronize()
    print('dw_layout %d  %.3f ms per step' % (lay, e0.elapsed_time(e1) / 20), flush=True)
    tr.close()
