"""Mid-size batches: persons/s of the fused mono pipeline per route (256x256-tile kernels / dense_mid_kernel with 64- and
128-row tiles) over a row sweep.  python tools/mid_sweep.py [rows ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import synth  # noqa: E402
from monoloco_amd import engine  # noqa: E402


def main():
    rows = [int(a) for a in sys.argv[1:]] or [2048, 3072, 4096, 6144, 8192, 12288, 16384, 24576, 32768]
    dev = torch.device('cuda:0')
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
    eng = engine.LocoEngine(sd, device=dev)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    routes = {'small/tile': dict(mid_rows=0, small_rows=2048), 'mid64': dict(mid_rows=1 << 30, mid_tile=64, small_rows=0),
              'mid128': dict(mid_rows=1 << 30, mid_tile=128, small_rows=0), 'half (w4 256x128)': dict(mid_rows=1 << 30, mid_tile=256, small_rows=0)}
    print('%8s ' % 'rows' + ' '.join('%22s' % r for r in routes))
    for m in rows:
        kps = torch.tensor(synth.make_poses(m, 3)).to(dev)
        out = torch.empty((m, 16), dtype=torch.float32, device=dev)
        xyzds = torch.empty((m, 5), dtype=torch.float32, device=dev)
        line = '%8d ' % m
        for name, kw in routes.items():
            eng.set_tuning(**kw)
            for _ in range(5):
                eng.forward_mono(kps, kinv, out=out, xyzds=xyzds)
            torch.cuda.synchronize()
            best = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 30
                e0.record()
                for _ in range(n):
                    eng.forward_mono(kps, kinv, out=out, xyzds=xyzds)
                e1.record()
                torch.cuda.synchronize()
                best.append(e0.elapsed_time(e1) / n)
            ms = float(np.median(best))
            line += '%9.1f us %7.2f M/s ' % (ms * 1e3, m / ms / 1e3)
        print(line, flush=True)
    eng.close()


if __name__ == '__main__':
    main()
