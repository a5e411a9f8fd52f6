#!/usr/bin/env python3
"""Training-step timing (BASELINE config 5): LocoModel 34->1024->9 fwd+bwd+MultiTaskLoss+clip+Adam on one MI355X,
(a) the reference fixture batch (331 rows), (b) a synthetic 65536-row batch; next to the CPU oracle
(oracle/train_oracle.py = torch-CPU autograd restatement of the reference loop) on the same batches.
Prints one JSON line per batch size."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, synth
from monoloco_amd.train import HipTrainer
from oracle.train_oracle import OracleTrainer

ap = argparse.ArgumentParser(); ap.add_argument('--steps', type=int, default=10); ap.add_argument('--cpu-seconds', type=float, default=10)
args = ap.parse_args()
dev = torch.device('cuda', 0)
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_train_inputs.npz')))
fx, fy = torch.tensor(g['mono_x']), torch.tensor(g['mono_y'])
rng = np.random.default_rng(0)
idx = rng.integers(0, len(fx), 65536)
bx = fx[idx] + torch.tensor(rng.normal(0, 0.01, (65536, 34)).astype(np.float32))
by = fy[idx]
for name, x, y in (('fixture-331', fx, fy), ('synthetic-65536', bx, by)):
    tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=dev)
    xd, yd = x.to(dev), y.to(dev)
    for _ in range(3): tr.step(xd, yd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps): tr.step(xd, yd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    tr.close()
    o = OracleTrainer(sd, lr=0.001, p_dropout=0.2)
    o.step(x[:512], y[:512])
    reps, tc = 0, 0.0
    while tc < args.cpu_seconds and reps < 20:
        t0 = time.perf_counter(); o.step(x, y); tc += time.perf_counter() - t0; reps += 1
    flop = 3 * 16865280 * len(x)   # fwd + 2x bwd, algorithmic
    print(json.dumps({"workload": "training step " + name + ", LocoModel 34->1024->9, dropout 0.2, fp32 (rows >= 4096: forward and data-gradient GEMMs of the hidden layers on 3-product fp16 MFMA, the rest exact-fp32 MFMA)",
                      "rows": len(x), "ms_per_step": round(dt * 1e3, 3), "rows_per_s": round(len(x) / dt, 1),
                      "algorithmic_tflops": round(flop / dt / 1e12, 2),
                      "cpu_baseline": {"ms_per_step": round(tc / reps * 1e3, 1), "rows_per_s": round(len(x) * reps / tc, 1),
                                       "cores": torch.get_num_threads(), "kind": "port"}}), flush=True)
