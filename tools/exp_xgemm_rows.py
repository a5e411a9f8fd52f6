"""Round 6: us per launch of the mid route's fp32-MFMA GEMM (xgemm_kernel<0, 0>, the forward product z = a . W^T, N = K = 1024) against the
number of rows -- 176 workgroups at 331 rows, one per CU at 512, two / four per CU at 1024 / 2048: what co-residency buys this kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch
from monoloco_amd import _lib
from monoloco_amd._lib import check
from monoloco_amd.engine import _ptr, _stream
dev = torch.device('cuda', 0)
lib = _lib.load()
N = K = 1024
b = torch.randn(N, K, device=dev)
for M in (176, 331, 512, 768, 1024, 1536, 2048, 4096):
    a = torch.randn(M, K, device=dev)
    c = torch.empty(M, N, device=dev)
    def run():
        check(lib.ml_debug_xgemm(_ptr(a), K, 0, _ptr(b), K, 0, _ptr(c), M, N, K, None, None, None, _stream(dev)), train=True)
    for _ in range(50): run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): run()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 300 * 1e6
    wgs = (N // 64) * ((M + 31) // 32)
    print("rows %5d  workgroups %4d (%.2f per CU)  %.1f us per launch  %.1f TFLOP/s  us per 512-row equivalent %.1f" % (
        M, wgs, wgs / 256, us, 2.0 * M * N * K / us / 1e6, us * 512 / M), flush=True)
