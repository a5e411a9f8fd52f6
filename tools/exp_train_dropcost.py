"""What the dropout hash costs the element-wise passes of the large-batch training step: exp_train_dropcost.py [rows] [p ...]
(ms per step by HIP events; run under rocprofv3 --kernel-trace --stats with ONE p for the per-kernel picture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ps = [float(a) for a in sys.argv[2:]] or [0.2, 0.0, 0.2, 0.0]
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
rng = np.random.default_rng(0)
idx = rng.integers(0, len(g['mono_x']), M)
x = (torch.tensor(g['mono_x'])[idx] + torch.tensor(rng.normal(0, 0.01, (M, 34)).astype(np.float32))).to(dev)
y = torch.tensor(g['mono_y'])[idx].to(dev)
for p in ps:
    tr = HipTrainer(sd, p_dropout=p, lr=0.001, device=dev)
    for _ in range(5):
        tr.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        tr.step(x, y)
    e1.record()
    torch.cuda.synchronize()
    print('rows %d  p_dropout %.2f  %.3f ms per step' % (M, p, e0.elapsed_time(e1) / n), flush=True)
    tr.close()
