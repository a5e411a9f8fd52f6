"""Round 6: timing ablations of dense_mid_kernel's LDS-DMA loop (builds with -DML_MID_ABL=<bits>: `make OUT=../lib/libmonoloco_hip_abl<bits>.so
EXTRA=-DML_MID_ABL=<bits>`; results are garbage, only the time is read): each build in its own process (MONOLOCO_HIP_LIB), us per forward of
the mono pipeline at the given row counts.  bits: 1 no requests in the loop, 2 no MFMAs, 4 no fragment reads, 8 no barrier / wait, 16 no stores."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path[:0] = [%r, %r]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
sd = synth.make_state_dict(1, 34, 9, 1024)
eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=dev, reserve_rows=16384)
eng.set_option('mid_splitk', 1)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
out = []
for m in [int(a) for a in sys.argv[1:]]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    o = torch.empty((m, 16), device=dev); x = torch.empty((m, 5), device=dev)
    for _ in range(300): eng.forward_mono(kps, kinv, out=o, xyzds=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): eng.forward_mono(kps, kinv, out=o, xyzds=x)
    torch.cuda.synchronize()
    out.append('%%d rows %%.1f us' %% (m, (time.perf_counter() - t0) / 400 * 1e6))
print('   '.join(out))
''' % (ROOT, os.path.join(ROOT, 'tests'))
rows = sys.argv[1:] or ['2048', '4096']
for bits in ('', '1', '2', '4', '8', '16', '3', '19'):
    lib = os.path.join(ROOT, 'monoloco_amd', 'lib', 'libmonoloco_hip%s.so' % ('_abl' + bits if bits else ''))
    if not os.path.exists(lib):
        continue
    env = dict(os.environ, MONOLOCO_HIP_LIB=lib)
    r = subprocess.run([sys.executable, '-c', CHILD] + rows, capture_output=True, text=True, env=env)
    print('ML_MID_ABL=%-3s %s' % (bits or '0', (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
