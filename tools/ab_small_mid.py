"""Round 6: where should the small-row kernels hand over to dense_mid_kernel?  us per forward of the mono pipeline at 128 .. 768 rows with the
small-row path (small_rows = 512, rounds 1-5) and with the mid path from 129 rows on (LDS-DMA loader, split-K filling the idle CUs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in [int(a) for a in (sys.argv[1:] or ['128', '192', '256', '320', '384', '448', '512', '640', '768'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    res = {}
    for name, small, sk in (('small path', 512, 1), ('mid, 1 range', 128, 1), ('mid, 2 ranges', 128, 2), ('mid, 4 ranges', 128, 4), ('mid, auto', 128, -1)):
        eng.set_tuning(small_rows=small)
        eng.set_option('mid_splitk', sk)
        out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev); raw = torch.empty((m, 9), device=dev)
        for _ in range(300):
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 400 * 1e6
        eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds, raw=raw)
        torch.cuda.synchronize()
        res[name] = (us, raw.clone(), eng.route_for_rows(m))
    base = res['small path'][1]
    print("rows %4d  " % m + "   ".join("%s (%s) %.1f us [%.1e]" % (k, v[2], v[0], (v[1] - base).abs().max().item()) for k, v in res.items()), flush=True)
