"""Round 6: A/B of an environment switch of the trainer on the large-batch step: exp_train_env_ab.py NAME [rows]
(one dropout-0 step with NAME=0 and NAME=1: every gradient compared; then ms per step alternating)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
NAME = sys.argv[1]
M = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
rng = np.random.default_rng(0)
idx = rng.integers(0, len(g['mono_x']), M)
x = (torch.tensor(g['mono_x'])[idx] + torch.tensor(rng.normal(0, 0.01, (M, 34)).astype(np.float32))).to(dev)
y = torch.tensor(g['mono_y'])[idx].to(dev)
res = {}
for v in ('0', '1'):
    os.environ[NAME] = v
    tr = HipTrainer(sd, p_dropout=0.0, lr=0.001, device=dev)
    losses, outs = tr.step(x, y, update=False, want_outputs=True)
    res[v] = (losses, outs.clone(), {k: t_.clone() for k, t_ in tr.grads().items()})
    tr.close()
same = all(torch.equal(res['0'][2][k], res['1'][2][k]) for k in res['0'][2]) and torch.equal(res['0'][1], res['1'][1])
worst = max(((res['0'][2][k].double() - res['1'][2][k].double()).norm() / (res['0'][2][k].double().norm() + 1e-30)).item() for k in res['0'][2])
print('%s 0 vs 1: same bits everywhere: %s   worst rms-rel gradient difference %.3e' % (NAME, same, worst))
for v in ('0', '1', '0', '1'):
    os.environ[NAME] = v
    tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev)
    for _ in range(5): tr.step(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): tr.step(x, y)
    e1.record(); torch.cuda.synchronize()
    print('%s=%s  %.3f ms per step' % (NAME, v, e0.elapsed_time(e1) / 20), flush=True)
    tr.close()
