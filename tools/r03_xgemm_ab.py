"""Timing ablations of the mid route's exact-fp32 GEMM (ml_debug_xgemm, flags >> 8 = xgemm_kernel's ABL template parameter):
which part of a k-step costs what.  Launches are issued from Python (~7 us each): durations below that are a floor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from monoloco_amd import _lib
from monoloco_amd._lib import check
from monoloco_amd.engine import _ptr, _stream
lib = _lib.load()
dev = torch.device('cuda', 0)
NAMES = {0: 'full', 1: 'two accumulators', 2: 'no MFMAs', 4: 'no global loads', 8: 'no LDS traffic', 12: 'no loads, no LDS (MFMAs + barriers)',
         14: 'barriers only', 16: 'no barriers (garbage)', 30: 'empty loop'}
for (M, N, K) in ((331, 1024, 1024), (512, 1024, 1024), (331, 1024, 4096)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05
    c = torch.empty(M, N, device=dev)
    print('forward layout, M %d N %d K %d (%d workgroups, %d k-steps)' % (M, N, K, (N // 64) * ((M + 31) // 32), K // 32))
    for abl in (0, 1, 2, 4, 8, 12, 14, 16, 30):
        def run(n):
            for _ in range(n):
                check(lib.ml_debug_xgemm(_ptr(a), K, 0, _ptr(b), K, 0, _ptr(c), M, N, K, None, None, None, abl << 8, _stream(dev)), train=True)
        run(20); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(200); e1.record(); torch.cuda.synchronize()
        print('   ABL %2d %-40s %.2f us per launch' % (abl, NAMES[abl], e0.elapsed_time(e1) / 200 * 1e3), flush=True)
