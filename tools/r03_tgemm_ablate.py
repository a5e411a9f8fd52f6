"""Timing ablations of the mid route's GEMM (ml_debug_tgemm, TGemmParams::dbg): which part of a k-step costs what."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from monoloco_amd import _lib
from monoloco_amd._lib import check
from monoloco_amd.engine import _ptr, _stream
lib = _lib.load()
dev = torch.device('cuda', 0)
NAMES = {0: 'full', 1: 'no loads', 2: 'no conversion/LDS stores', 4: 'no reads/MFMAs', 3: 'no loads, no stores', 6: 'no stores, no MFMAs (loads + barriers)',
         7: 'barriers only', 15: 'empty loop', 8: 'no barriers (garbage)'}
for (M, N, K, tile) in ((331, 1024, 1024, 32), (512, 1024, 1024, 32), (1024, 1024, 352, 64), (1024, 1024, 1024, 64)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
    c = torch.empty(M, N, device=dev)
    bmax = b.abs().max().reshape(1).clone()
    print('M %d N %d K %d tile %d  (%d workgroups)' % (M, N, K, tile, (N // 64) * ((M + tile - 1) // tile)))
    for dbg in (0, 1, 2, 4, 3, 6, 7, 15, 8):
        def run(n):
            for _ in range(n):
                check(lib.ml_debug_tgemm(_ptr(a), _ptr(b), _ptr(c), M, N, K, None, None, None, _ptr(bmax), None, 0, tile | (dbg << 8), _stream(dev)), train=True)
        run(20); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(200); e1.record(); torch.cuda.synchronize()
        print('   dbg %2d %-42s %.2f us per launch' % (dbg, NAMES[dbg], e0.elapsed_time(e1) / 200 * 1e3), flush=True)
