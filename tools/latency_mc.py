"""Latency of Loco-style MC-dropout epistemic uncertainty for one image (16 persons, 50 passes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}, device=dev)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m, npass in ((16, 50), (16, 10), (256, 50), (65536, 4)):
    kps = torch.tensor(synth.make_keypoints(m, seed=1)).to(dev)
    for _ in range(5): eng.epistemic_mono(kps, kinv, npass)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): eng.epistemic_mono(kps, kinv, npass)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("epistemic: %6d persons x %3d passes: %.1f us/call (%.1f us per pass)" % (m, npass, dt * 1e6, dt * 1e6 / npass), flush=True)
