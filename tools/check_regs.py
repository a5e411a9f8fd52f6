"""Compile the library with -Rpass-analysis and print VGPR/SGPR/spill/scratch/LDS per kernel; exit 1 if a
dense kernel spills (spill stores are VMEM ops and would break the kernel's counted vmcnt waits)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'monoloco_amd', 'csrc', sys.argv[1] if len(sys.argv) > 1 else 'monoloco_hip.hip')
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'], capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r'remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s+(\S+) \[-Rpass', line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
bad = 0
for k, v in rows.items():
    name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip()[:70]
    spill = int(v.get('VGPRs Spill', 0)) + int(v.get('ScratchSize [bytes/lane]', 0))
    flag = ''
    if 'dense_kernel' in k and spill:
        flag = '  <-- SPILL'; bad = 1
    print("%-72s VGPR %-4s AGPR %-4s occ %-2s SGPR %-4s vspill %-3s sspill %-3s scratch %-4s LDS %s%s" % (name, v.get('VGPRs'), v.get('AGPRs'), v.get('Occupancy [waves/SIMD]'), v.get('TotalSGPRs'),
          v.get('VGPRs Spill'), v.get('SGPRs Spill'), v.get('ScratchSize [bytes/lane]'), v.get('LDS Size [bytes/block]'), flag))
sys.exit(bad)
