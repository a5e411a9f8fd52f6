"""Round 6 A/B inside one process: the mono pipeline in the mid window with the input layer pre-processing its own persons
(dense_mid_kernel<.., PREP>, `mid_prep` 1) against prep_kernel in front of it (`mid_prep` 0); alternating blocks, outputs compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in [int(a) for a in (sys.argv[1:] or ['512', '1024', '2048', '3072', '4096', '6144', '8192'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    res, outs = {0: [], 1: []}, {}
    for rep in range(3):
        for fused in (0, 1):
            eng.set_option('mid_prep', fused)
            out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev)
            for _ in range(300):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(400):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            res[fused].append((time.perf_counter() - t0) / 400 * 1e6)
            outs[fused] = out.clone()
    print("rows %5d (%s)  prep_kernel + layer %s us   fused %s us   (%.2f -> %.2f M persons/s)  same bits: %s" % (
        m, eng.route_for_rows(m), ['%.1f' % v for v in res[0]], ['%.1f' % v for v in res[1]], m / min(res[0]), m / min(res[1]),
        bool(torch.equal(outs[0], outs[1]))), flush=True)
