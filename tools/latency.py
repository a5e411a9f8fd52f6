"""Per-call latency of the device pipeline at small row counts (one image's worth of persons)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=8192)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in [int(a) for a in (sys.argv[1:] or ['1', '16', '64', '256', '512', '1024', '2048', '4096'])]:
    kps = torch.tensor(synth.make_keypoints(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev)
    for _ in range(300):  # long enough for the clocks to settle at this load
        eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("rows %5d  SMALL_ROWS=%s: %.1f us/call  (%.2f M persons/s)" % (m, os.environ.get('ML_SMALL_ROWS', 'default'), dt * 1e6, m / dt / 1e6), flush=True)
