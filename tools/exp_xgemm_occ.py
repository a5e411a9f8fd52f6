"""xgemm at more than two workgroups per CU: which 32 x 64 tiles of a 2048 x 1024 x 1024 product (both operands k-contiguous) are
wrong, and, per wrong tile, the projection of its error on every k32 step's own contribution (-1 = that step is missing, +1 =
counted twice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import ctypes
import numpy as np, torch
from monoloco_amd import _lib
from monoloco_amd._lib import check
from monoloco_amd.engine import _ptr, _stream

dev = torch.device('cuda', 0)
lib = _lib.load()
M, N, K = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (2048, 1024, 1024)
gen = torch.Generator().manual_seed(5)
A = torch.randn(M, K, generator=gen).to(dev)
B = (torch.randn(N, K, generator=gen) * 0.05).to(dev)
C = torch.empty(M, N, device=dev)
for rep in range(int(os.environ.get('REPS', '2'))):
    C.fill_(float('nan'))
    with torch.cuda.device(dev):
        check(lib.ml_debug_xgemm(_ptr(A), K, 0, _ptr(B), K, 0, _ptr(C), M, N, K, None, None, None, _stream(dev)), train=True)
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t()
    D = (C.double() - ref)
    tile_err = D.abs().view(M // 32, 32, N // 64, 64).amax(dim=(1, 3)).cpu().numpy()     # (row tiles, column tiles)
    bad = np.argwhere(tile_err > 1e-4)
    print('run %d: %d of %d tiles wrong; workgroup ids (by * %d + bx): %s' % (rep, len(bad), tile_err.size, N // 64,
          ' '.join(str(int(by * (N // 64) + bx)) for by, bx in bad[:400])))
    rows_bad = sorted(set(int(b[0]) for b in bad))
    print('   row tiles with a wrong tile:', rows_bad)
    for by, bx in bad[:(6 if rep < 2 else 0)]:
        Dt = D[by * 32:(by + 1) * 32, bx * 64:(bx + 1) * 64]
        coef = []
        for k in range(K // 32):
            Pk = A[by * 32:(by + 1) * 32, k * 32:(k + 1) * 32].double() @ B[bx * 64:(bx + 1) * 64, k * 32:(k + 1) * 32].double().t()
            coef.append(float((Dt * Pk).sum() / (Pk * Pk).sum()))
        # which rows / columns of the tile carry the error
        print('   tile (%d, %d): max err %.3f; rows with error %d/32, columns %d/64; projection on the k-steps: %s' % (
            by, bx, Dt.abs().max().item(), int((Dt.abs().amax(1) > 1e-4).sum()), int((Dt.abs().amax(0) > 1e-4).sum()),
            ' '.join('%+.2f' % c for c in coef)))
