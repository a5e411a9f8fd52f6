"""ms per training step on small batches (the reference's regime: batch 64 ... 1024), hidden 1024."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
out = []
for m in [int(a) for a in (sys.argv[1:] or ['64', '331', '512', '1024'])]:
    xb, yb = synth.big_train_batch(g['mono_x'], g['mono_y'], m, 3)
    x, y = torch.tensor(xb).to(dev), torch.tensor(yb).to(dev)
    tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev)
    for _ in range(5): tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): tr.step(x, y)
    torch.cuda.synchronize(); out.append("%d: %.3f" % (m, (time.perf_counter() - t0) / 40 * 1e3))
    tr.close()
print("ms per step by rows  " + "  ".join(out), flush=True)
