"""Same-box A/B of library builds on the large-batch training step: lib_ab_train.py <lib.so> <lib.so> ... [-- rows]
(one process per build and round through MONOLOCO_HIP_LIB, alternating, ms per step from tools/exp_train_dropcost.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rows = '65536'
if '--' in args:
    k = args.index('--'); rows = args[k + 1]; args = args[:k]
res = {a: [] for a in args}
for rnd in range(3):
    for lib in args:
        env = dict(os.environ, MONOLOCO_HIP_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'exp_train_dropcost.py'), rows, '0.2'], env=env,
                             capture_output=True, text=True).stdout
        ms = [float(l.split()[-4]) for l in out.splitlines() if 'ms per step' in l]
        res[lib].append(ms[0] if ms else float('nan'))
for lib in args:
    print('%-50s %s ms per step at %s rows' % (os.path.basename(lib), ['%.3f' % v for v in res[lib]], rows), flush=True)
