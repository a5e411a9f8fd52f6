"""Same-box A/B of several builds of the library (each in its own process, MONOLOCO_HIP_LIB), two alternating rounds: us per forward of the
mono pipeline at the given row counts.   python tools/lib_ab.py <lib.so> [<lib.so> ...] -- <rows> [<rows> ...]"""
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
kinv = engine.inverse_intrinsics(synth.KITTI_K)
out = []
for m in [int(a) for a in sys.argv[1:]]:
    kps = torch.tensor(synth.make_poses(m, seed=1)).to(dev)
    o = torch.empty((m, 16), device=dev); x = torch.empty((m, 5), device=dev)
    for _ in range(400): eng.forward_mono(kps, kinv, out=o, xyzds=x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(600): eng.forward_mono(kps, kinv, out=o, xyzds=x)
    torch.cuda.synchronize()
    out.append('%%d rows %%.1f us' %% (m, (time.perf_counter() - t0) / 600 * 1e6))
print('   '.join(out))
''' % (ROOT, os.path.join(ROOT, 'tests'))
args = sys.argv[1:]
libs, rows = args[:args.index('--')], args[args.index('--') + 1:]
for rnd in (1, 2):
    for lib in libs:
        path = lib if os.path.isabs(lib) else os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, '-c', CHILD] + rows, capture_output=True, text=True, env=dict(os.environ, MONOLOCO_HIP_LIB=path))
        print('%-44s %s' % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]), flush=True)
