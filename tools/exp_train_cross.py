"""Where does each route of the training GEMMs pay?  ms per step: exact-fp32 route, mid route (train_mid.h), large-batch route."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
dev = torch.device('cuda', 0)
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
for m in (64, 128, 256, 331, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    xb, yb = synth.big_train_batch(g['mono_x'], g['mono_y'], m, 3)
    x, y = torch.tensor(xb).to(dev), torch.tensor(yb).to(dev)
    res = {}
    for name in ('exact', 'mid', 'fast'):
        if name == 'exact' and m > 8192:
            res[name] = float('nan')
            continue
        tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev, route=name)
        for _ in range(3): tr.step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): tr.step(x, y)
        torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
        tr.close()
    print("rows %6d: exact %.3f ms, mid %.3f ms, fast %.3f ms" % (m, res['exact'], res['mid'], res['fast']), flush=True)
