"""Round 5 A/B inside one process: the mid window (513 .. 8192 rows) with both heads riding in the dense epilogues (dense_mid_kernel's /
the half-size w4 tile's) + tail_mono_kernel (mid_heads 1, the default) against heads_pair_kernel behind the last layer (mid_heads 0, rounds 3-4); alternating
blocks on one engine, the two routes' outputs compared.  Optional arguments: row counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
for m in [int(a) for a in (sys.argv[1:] or ['1024', '2048', '3072', '4096', '6144', '8192'])]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    conf = torch.rand(m, device=dev)
    outs = {}
    res = {0: [], 1: []}
    for rep in range(3):
        for hh in (0, 1):
            eng.set_option('mid_heads', hh)
            out = torch.empty((m, 16), device=dev); xyzds = torch.empty((m, 5), device=dev); raw = torch.empty((m, 9), device=dev)
            for _ in range(200):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            n = 300
            t0 = time.perf_counter()
            for _ in range(n):
                eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            res[hh].append((time.perf_counter() - t0) / n * 1e6)
            eng.forward_mono(kps, kinv, box_conf=conf, out=out, xyzds=xyzds, raw=raw)
            torch.cuda.synchronize()
            outs[hh] = (out.clone(), xyzds.clone(), raw.clone())
    d_raw = (outs[0][2] - outs[1][2]).abs().max().item()
    d_xyz = (outs[0][1] - outs[1][1]).abs().max().item()
    print("rows %5d  plan %s" % (m, eng.plan_for_rows(m)))
    print("rows %5d  pair kernel %s us   fused heads %s us   (%.2f -> %.2f M persons/s)   |raw| apart %.2e  |xyzds| apart %.2e" % (
        m, ['%.1f' % v for v in res[0]], ['%.1f' % v for v in res[1]], m / min(res[0]), m / min(res[1]), d_raw, d_xyz), flush=True)
