"""Round 6: would the mid route's exact-fp32 MFMA GEMM carry the large-batch route's INPUT layer forward (z0 = x . W1^T, 65536 x 34 -> 1024,
skinny_out_kernel: 160 us = 1.7 TB/s of its 268 MB output)?  x padded to K = 64."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from monoloco_amd import _lib
from monoloco_amd._lib import check
from monoloco_amd.engine import _ptr, _stream
dev = torch.device('cuda', 0)
lib = _lib.load()
M, N = 65536, 1024
for K in (32, 64):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    c = torch.empty(M, N, device=dev)
    def run():
        check(lib.ml_debug_xgemm(_ptr(a), K, 0, _ptr(b), K, 0, _ptr(c), M, N, K, _ptr(bias), None, None, _stream(dev)), train=True)
    for _ in range(20): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): run()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 100 * 1e6
    ref = a.double() @ b.double().T + bias.double()
    print("K %d: %.1f us per launch (%.2f TB/s of output)  max |c - fp64| %.2e" % (K, us, M * N * 4 / us / 1e6, (c.double() - ref).abs().max().item()), flush=True)
