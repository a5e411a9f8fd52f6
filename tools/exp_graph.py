"""Is the single-image pipeline host-launch-bound?  Replay it from a captured graph and compare with eager launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=8192)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in (1, 16, 64, 256):
    kps = torch.tensor(synth.make_keypoints(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(20): eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
    torch.cuda.synchronize()
    ref = out.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
    out.zero_()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n): eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay(); torch.cuda.synchronize()
    tgs = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds); torch.cuda.synchronize()
    tes = (time.perf_counter() - t0) / n
    print("rows %4d: eager %.1f us back-to-back, %.1f us synchronous | graph replay %.1f us back-to-back, %.1f us synchronous" % (m, te * 1e6, tes * 1e6, tg * 1e6, tgs * 1e6), flush=True)
