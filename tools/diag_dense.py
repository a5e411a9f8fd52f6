"""GPU diagnostic: probe the dense kernel with one-hot operands and report where the product lands.
Only used while bringing the kernel up (prints, asserts nothing)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from monoloco_amd import engine

dev = torch.device('cuda', 0)
m, k, n = 256, 64, 256
bad = 0
for (mi, ki, ni) in [(0, 0, 0), (1, 0, 0), (0, 0, 1), (0, 1, 0), (5, 9, 37), (37, 40, 5), (200, 63, 130), (130, 17, 200), (255, 33, 255)]:
    x = np.zeros((m, k), np.float32); x[mi, ki] = 3.0
    w = np.zeros((n, k), np.float32); w[ni, ki] = 2.0
    y = engine.debug_linear(torch.tensor(x, device=dev), w, np.zeros(n, np.float32)).cpu().numpy()
    nz = np.argwhere(y != 0)
    ok = len(nz) == 1 and tuple(nz[0]) == (mi, ni) and y[mi, ni] == 6.0
    bad += (not ok)
    print("probe x[%d,%d] w[%d,%d]: nonzeros %s values %s %s" % (mi, ki, ni, ki, nz[:6].tolist(), y[tuple(nz[:6].T)].tolist() if len(nz) else [], "OK" if ok else "MISMATCH"))
# k mismatch probe: x one-hot at k=ki, w one-hot at different k -> must be all zero
x = np.zeros((m, k), np.float32); x[3, 7] = 1
w = np.zeros((n, k), np.float32); w[4, 8] = 1
y = engine.debug_linear(torch.tensor(x, device=dev), w, np.zeros(n, np.float32)).cpu().numpy()
print("k-mismatch probe nonzeros:", np.argwhere(y != 0)[:6].tolist())
# bias / relu / residual probes
x = np.zeros((m, k), np.float32); w = np.zeros((n, k), np.float32)
b = np.arange(n, dtype=np.float32) - 100
r = (np.arange(m * n, dtype=np.float32).reshape(m, n) % 97) * 0.25
y = engine.debug_linear(torch.tensor(x, device=dev), w, b, relu=True, res=torch.tensor(r, device=dev)).cpu().numpy()
ref = np.maximum(b, 0)[None, :] + r
print("bias+relu+res max err", np.abs(y - ref).max())
print("SUMMARY bad probes:", bad)
