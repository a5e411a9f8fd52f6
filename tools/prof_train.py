"""Training steps only (for rocprofv3 --kernel-trace --stats): prof_train.py [steps] [rows, default 65536; 331 = the fixture batch]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
rng = np.random.default_rng(0)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
idx = rng.integers(0, len(g['mono_x']), M) if M != 331 else np.arange(331)
x = (torch.tensor(g['mono_x'])[idx] + torch.tensor((rng.normal(0, 0.01, (M, 34)) * (0 if M == 331 else 1)).astype(np.float32))).to(dev)
y = torch.tensor(g['mono_y'])[idx].to(dev)
tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    tr.step(x, y)
torch.cuda.synchronize()
tr.close()
