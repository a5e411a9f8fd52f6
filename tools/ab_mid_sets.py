"""Round 5 A/B inside one process: dense_mid_kernel's 64-row tile with two (rounds 3-4) and three loader register sets (requests
two / three k-steps ahead of their LDS store), alternating blocks on one engine; the arithmetic is the same, so the outputs must be
bit-identical.  Optional arguments: row counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in [int(a) for a in (sys.argv[1:] or ['600', '1024', '1536', '2048', '3072', '4000'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    outs = {}
    res = {2: [], 3: []}
    for rep in range(3):
        for ns in (2, 3):
            eng.set_option('mid_sets', ns)
            out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev); raw = torch.empty((m, 9), device=dev)
            for _ in range(200):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            res[ns].append((time.perf_counter() - t0) / n * 1e6)
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds, raw=raw)
            torch.cuda.synchronize()
            outs[ns] = raw.clone()
    print("rows %5d  2 sets %s us   3 sets %s us   (%.2f -> %.2f M persons/s)   same bits: %s" % (
        m, ['%.1f' % v for v in res[2]], ['%.1f' % v for v in res[3]], m / min(res[2]), m / min(res[3]), torch.equal(outs[2], outs[3])), flush=True)
eng.set_option('mid_sets', 0)
