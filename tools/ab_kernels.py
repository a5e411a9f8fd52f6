"""Within-process interleaved A/B of the dense-kernel selections at the headline batch (cdna_hip_programming.md rule 24):
pp everywhere / default mix / w4 everywhere / dense_mid_kernel with 128 x 128 tiles (two co-resident workgroups per CU) / with
128 x 64 tiles (three), R rounds x 10 steps each, median and min per variant."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}
kinv = engine.inverse_intrinsics(synth.KITTI_K)
M = 65536
kps = torch.tensor(synth.make_keypoints(M, seed=100)).to(dev)
conf = torch.rand(M, device=dev)
eng = engine.LocoEngine(sd, device=dev, reserve_rows=M)
out = torch.empty((M, 16), device=dev); xyzds = torch.empty((M, 5), device=dev)
variants = {'pp': (2, False, 0), 'mix': (4, False, 0), 'w4': (4, True, 0), 'mid128': (4, False, 128), 'mid64': (4, False, 64)}
res = {k: [] for k in variants}
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    for name, (k, allw, mid) in variants.items():
        eng.set_tuning(tile_kernel=k, everywhere=allw, mid_rows=(1 << 30) if mid else 0, mid_tile=mid)
        for _ in range(3):
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 10 * 1e3)
for name, v in res.items():
    print("%-6s median %.4f ms  min %.4f  max %.4f  (%.2f M persons/s at the median)" % (name, statistics.median(v), min(v), max(v), M / statistics.median(v) / 1e3))
