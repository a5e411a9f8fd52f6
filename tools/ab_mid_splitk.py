"""Round 6 A/B inside one process: the mid window with dense_mid_kernel's reduction cut into k ranges (split-K across workgroups, the last
arriver of a tile runs the epilogue: `mid_splitk` -1 auto / 2 / 4) against one workgroup per tile (`mid_splitk` 1, rounds 3-5), per tile
height, alternating blocks on one engine; the routes' outputs compared.  Optional arguments: row counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
variants = [(0, 1, 0), (0, 1, 1), (64, 1, 1), (128, 1, 1), (64, 2, 1), (128, 2, 1), (64, 4, 1), (0, -1, 1)]   # (mid_tile, mid_splitk, mid_dma)
if os.environ.get('AB_WGS'):
    eng.set_option('mid_wgs', int(os.environ['AB_WGS']))
for m in [int(a) for a in (sys.argv[1:] or ['1024', '2048', '3072', '4096', '6144', '8192'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    eng.set_tuning(mid_tile=64 if m <= 4096 else 128)   # keep the whole window on dense_mid_kernel (no half-size w4 tile) for the A/B
    outs, res = {}, {v: [] for v in variants}
    for rep in range(3):
        for v in variants:
            tile, sk, dma = v
            eng.set_tuning(mid_tile=tile if tile else (64 if m < 4096 else 128))
            eng.set_option('mid_splitk', sk)
            eng.set_option('mid_dma', dma)
            out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev); raw = torch.empty((m, 9), device=dev)
            for _ in range(200):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            res[v].append((time.perf_counter() - t0) / n * 1e6)
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds, raw=raw)
            torch.cuda.synchronize()
            outs[v] = raw.clone()
    base = outs[variants[0]]
    for v in variants:
        print("rows %5d  tile %3d splitk %2d dma %d: %s us  -> %.2f M persons/s   |raw| apart from the rounds-3-5 kernel %.2e" % (
            m, v[0], v[1], v[2], ['%.1f' % t for t in res[v]], m / min(res[v]), (outs[v] - base).abs().max().item()), flush=True)
