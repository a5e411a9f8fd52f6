"""Instruction mix of the hand-scheduled main loops, from the gfx950 assembly hipcc emits for csrc/monoloco_hip.hip.

The dense_kernel_w4 family issues its LDS-DMA through inline asm that hipcc's vmcnt bookkeeping cannot see and waits for it with
hand-counted `s_waitcnt vmcnt(N)`; a compiler-inserted vector-memory operation (a spill, a hoisted load) or a compiler-inserted
`s_waitcnt vmcnt(0)` inside the loop turns that into a silent data race or a stall.  This tool finds every kernel's main loop (the
innermost backward branch around the most MFMAs) and reports, per loop iteration (= two k32 steps): MFMAs, LDS-DMA instructions,
barriers, the multiset of vmcnt waits, LDS fragment reads, other vector-memory instructions and scratch accesses.
`python tools/check_loops.py [asm file]` prints the table; tests/test_build_guards.py asserts it (CPU, no GPU)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'monoloco_amd', 'csrc')
ASM_DIR = os.path.join(ROOT, 'monoloco_amd', 'lib', 'asm')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '--offload-arch=gfx950', '-Wall', '-Wno-unused-function']   # = csrc/Makefile CXXFLAGS


def sources_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith(('.h', '.hip')))


def build_asm(unit='monoloco_hip', extra=()):
    """Device-only assembly + resource remarks of one translation unit, cached under monoloco_amd/lib/asm/ until a source changes."""
    os.makedirs(ASM_DIR, exist_ok=True)
    tag = unit + ('_' + re.sub(r'[^A-Za-z0-9]+', '_', ' '.join(extra)) if extra else '')
    s_path, r_path = os.path.join(ASM_DIR, tag + '.s'), os.path.join(ASM_DIR, tag + '.remarks')
    if not (os.path.exists(s_path) and os.path.exists(r_path) and os.path.getmtime(s_path) >= sources_mtime()):
        hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
        cmd = [hipcc] + FLAGS + list(extra) + ['--cuda-device-only', '-S', os.path.join(CSRC, unit + '.hip'), '-o', s_path + '.tmp',
                                               '-Rpass-analysis=kernel-resource-usage']
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc failed: %s" % res.stderr[-2000:])
        open(r_path, 'w').write(res.stderr)
        os.replace(s_path + '.tmp', s_path)
    return s_path, r_path


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def functions(s_path):
    """{mangled name: [instruction lines]} of every kernel-sized function in the assembly."""
    funcs, cur, name = {}, None, None
    for line in open(s_path):
        m = re.match(r'^(_Z\w+):\s', line)
        if m:
            name, cur = m.group(1), []
            funcs[name] = cur
            continue
        if cur is not None:
            if line.startswith('\t.end_amdhsa_kernel') or line.startswith('.Lfunc_end'):
                cur = None
                continue
            cur.append(line.rstrip('\n'))
    return funcs


def main_loop(lines):
    """(first, last) line index of the innermost loop holding the most v_mfma instructions, or None."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    best, best_n = None, 0
    for a, b in loops:
        n = sum(1 for l in lines[a:b] if 'v_mfma' in l)
        inner = any(a < a2 and b2 < b and sum(1 for l in lines[a2:b2] if 'v_mfma' in l) for a2, b2 in loops if (a2, b2) != (a, b))
        if n > best_n and not inner:
            best, best_n = (a, b), n
    return best


def request_loop(lines):
    """(first, last) of the innermost loop holding the most LDS-DMA instructions and NO MFMA (a request-only wave's loop), or None."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r'^\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', l) or re.match(r'^\s+s_branch\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    best, best_n = None, 0
    for a, b in loops:
        inner = any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in loops)
        n = sum(1 for x in lines[a:b] if 'global_load_lds' in x)
        if n > best_n and not inner and not any('v_mfma' in x for x in lines[a:b]):
            best, best_n = (a, b), n
    return best


def mix(lines, span):
    a, b = span
    c = collections.Counter()
    waits = collections.Counter()
    for l in lines[a:b + 1]:
        t = l.strip()
        if not t or t.startswith((';', '.')):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'):
            c['mfma'] += 1
        elif op.startswith('global_load_lds'):
            c['lds_dma'] += 1
        elif op == 's_barrier':
            c['barrier'] += 1
        elif op == 's_waitcnt':
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                waits[int(m.group(1))] += 1
        elif op.startswith(('ds_read', 'ds_load')):
            c['ds_read'] += 1
        elif op.startswith(('ds_write', 'ds_store')):
            c['ds_write'] += 1
        elif op.startswith(('global_load', 'global_store', 'buffer_load', 'buffer_store', 'flat_load', 'flat_store', 'global_atomic')):
            c['other_vmem'] += 1
        elif op.startswith('scratch_'):
            c['scratch'] += 1
    c = dict(c)
    c['vmcnt_waits'] = dict(sorted(waits.items()))
    return c


def resources(r_path):
    """{mangled name: {remark: value}} from -Rpass-analysis=kernel-resource-usage."""
    cur, rows = None, {}
    for line in open(r_path):
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s+(\S+) \[-Rpass', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    return rows


def report(unit='monoloco_hip', extra=(), pattern='dense_kernel_w4'):
    s_path, r_path = build_asm(unit, extra)
    funcs = functions(s_path)
    res = resources(r_path)
    names = [n for n in funcs if pattern in n]
    dm = demangle(names)
    out = {}
    for n in names:
        span = main_loop(funcs[n])
        rspan = request_loop(funcs[n])
        out[dm[n].replace('void mlk::', '').replace('(mlk::DenseParams)', '')] = {
            'loop': mix(funcs[n], span) if span else None, 'request_loop': mix(funcs[n], rspan) if rspan else None,
            'resources': res.get(n, {})}
    return out


if __name__ == '__main__':
    pat = sys.argv[1] if len(sys.argv) > 1 else 'dense_kernel_w4'
    for k, v in report(pattern=pat).items():
        r = v['resources']
        print('%-52s VGPR %-4s AGPR %-4s scratch %-3s spill %-3s | %s' % (k, r.get('VGPRs'), r.get('AGPRs'), r.get('ScratchSize [bytes/lane]'),
                                                                       r.get('VGPRs Spill'), v['loop']))
