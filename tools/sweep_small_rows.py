"""Round 5: the small-row window (<= 512 rows) per tile shape -- us per forward_raw (8 dense layers + heads, back to back) for the
16 x 16 tiles (small32_rows large) against the 32 x 32 tiles (small32_rows 0) and dense_small_multi_kernel (16 x 16 tiles, a
workgroup keeps its weight rows and walks several row tiles; > 64 rows), mono and stereo widths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
for in_f, out_f in ((34, 9), (68, 10)):
    sd = synth.make_state_dict(1, in_f, out_f, 1024)
    eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=1024)
    for m in (16, 32, 48, 64, 80, 100, 128, 160, 200, 256, 384, 512):
        x = torch.randn(m, in_f, device=dev)
        res = {}
        for name, s32, multi in (('t16', 100000, 0), ('t32', 0, 0), ('multi', 128, 1)):
            eng.set_tuning(small32_rows=s32)
            eng.set_option('small_multi', multi)
            out = torch.empty((m, out_f), device=dev)
            for _ in range(100):
                eng.forward_raw(x, out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                eng.forward_raw(x, out=out)
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 300 * 1e6
        print("in %2d rows %4d: 16x16 tiles %.1f us   32x32 tiles %.1f us   16x16, several row tiles per workgroup %.1f us" % (
            in_f, m, res['t16'], res['t32'], res['multi']), flush=True)
    eng.close()
