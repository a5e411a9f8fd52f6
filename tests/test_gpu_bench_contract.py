"""-m gpu: the driver's contract of bench.py, executed: ONE JSON line on stdout with the agreed keys, the roofline and parity blocks,
the ranks record -- on a short run (4096 persons, 3 steps, no extras, no CPU leg)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract(hip_lib, cuda_device):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '4096', '--no-extra',
                        '--cpu-seconds', '0'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'dtype', 'data', 'config', 'roofline', 'parity', 'ranks_seen', 'ranks'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True and d['scaling'] == 'weak'
    assert d['vs_baseline'] is None and d['data'] == 'synthetic' and d['unit'] == 'persons/s' and 'workload' in d['config']
    assert abs(d['value'] - 4096 / d['ms_per_step'] * 1e3) <= 2e-3 * d['value']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0 and 0 < rf['frac'] < 0.34
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) <= 1e-3 and rf['launches'] == 3 * 8
    assert rf['traffic'] is None and rf['traffic_source'].startswith('none')      # (the committed PMC passes cover the 65536-row line only)
    assert d['parity']['max_abs_xyzds'] <= d['parity']['tolerance'] == 1e-4
    assert d['ranks_seen'] == 1 and d['ranks'][0]['rank'] == 0 and 'uuid' in d['ranks'][0]['device'].lower() or 'pci' in d['ranks'][0]['device'].lower()


def test_bench_line_carries_live_counters(hip_lib, cuda_device):
    """The headline configuration: roofline.traffic / mfma_busy / hbm_gbps are measured by THIS run (child runs under rocprofv3,
    bench.live_counters) and agree with the passes committed under profiles/ (kept beside them as *_replayed)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '6', '--warmup', '2', '--no-extra', '--cpu-seconds', '0', '--no-parity'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, "exactly one line on stdout"
    rf = json.loads(r.stdout)['roofline']
    assert rf['counters_source'].startswith('live'), rf['counters_source']
    assert rf['traffic'] == rf['traffic_live'] and rf['traffic_source'].startswith('live')
    assert rf['algorithmic_bytes_per_launch'] <= rf['traffic_live'] <= 1.6 * rf['algorithmic_bytes_per_launch']
    assert 0.3 < rf['mfma_busy_live'] < 1.0 and 500 < rf['hbm_gbps_live'] < 8000
    assert 'dense_kernel_w4' in rf['dominant_kernel_live'] and 200 < rf['dominant_kernel_avg_us_live'] < 600
    if rf.get('traffic_replayed'):
        assert abs(rf['traffic_live'] / rf['traffic_replayed'] - 1) < 0.10
        assert abs(rf['mfma_busy_live'] / rf['mfma_busy_replayed'] - 1) < 0.10
