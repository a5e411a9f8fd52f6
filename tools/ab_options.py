"""A/B of two tuning-option sets inside one process on the mono pipeline (alternating blocks, outputs compared):
AB_A="mid_tile=0" AB_B="mid_tile=128,mid_splitk=2" python tools/ab_options.py 1536 2048 2560"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
sets = [[kv.split('=') for kv in os.environ.get(n, d).split(',') if kv] for n, d in (('AB_A', ''), ('AB_B', ''))]
for m in [int(a) for a in (sys.argv[1:] or ['1024', '2048', '4096'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    res, outs = {0: [], 1: []}, {}
    for rep in range(3):
        for side in (0, 1):
            for k, v in sets[side]:   # (name both sides' values of every option: nothing is undone behind your back)
                if k == 'mid_tile':
                    eng.set_tuning(mid_tile=int(v))
                else:
                    eng.set_option(k, int(v))
            out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev)
            for _ in range(300):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(400):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            res[side].append((time.perf_counter() - t0) / 400 * 1e6)
            outs[side] = xyzds.clone()
    d = (outs[0] - outs[1]).abs().max().item()
    print("rows %5d  A %s us   B %s us   (%.2f -> %.2f M persons/s)  max |A - B| on (x,y,z,d,s) %.2e" % (
        m, ['%.1f' % v for v in res[0]], ['%.1f' % v for v in res[1]], m / min(res[0]), m / min(res[1]), d), flush=True)
