"""-m gpu: the N > 1 code path on the REAL backend with the REAL engine, at world size 1 (one GPU per box):
`nccl` (= RCCL) process-group initialisation with device_id, the collective into slices of one pre-allocated tensor, and
the stream ordering between the engine's kernels (torch's current stream) and the collective.  Scaling itself is
unmeasured here: the driver's 8-GPU run is the only place a curve can come from."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r'''
import os, sys
sys.path[:0] = [%r, %r]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29631', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch, torch.distributed as dist
import synth
from monoloco_amd import engine, parallel
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)     # what parallel.init_from_env does for N > 1
assert dist.get_backend() == 'nccl'
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}
eng = engine.LocoEngine(sd, device=dev, reserve_rows=8192)
kinv = engine.inverse_intrinsics(synth.KITTI_K)
m = 5000
for mode in ('gather', 'all_gather'):
    sharded = parallel.ShardedRows(m, 5, dev, mode=mode)
    assert (sharded.lo, sharded.hi) == (0, m)
    xyzds = torch.empty((m, 5), dtype=torch.float32, device=dev)
    for step in range(4):      # new inputs every step: a collective that ran ahead of post_kernel would ship stale rows
        kps = torch.tensor(synth.make_keypoints(m, seed=50 + step)).to(dev)
        def local(lo, hi):
            eng.forward_mono(kps[lo:hi], kinv, out=None, xyzds=xyzds)
            return xyzds
        full = sharded.run(local)
        torch.cuda.synchronize()
        assert full.data_ptr() == sharded.gather.full.data_ptr()          # pre-allocated, not re-created per step
        ref = eng.forward_mono(kps, kinv)[1]
        torch.cuda.synchronize()
        assert torch.equal(full, ref), (mode, step)
dist.barrier()
dist.destroy_process_group()
eng.close()
print('ok')
'''


def test_sharded_rows_on_nccl_world1(hip_lib, cuda_device):
    code = _CODE % (ROOT, os.path.join(ROOT, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, cwd=ROOT)
    # (RCCL prints its version banner to stdout at teardown)
    assert r.returncode == 0 and 'ok' in r.stdout.split(), r.stdout[-2000:] + r.stderr[-4000:]


def _bench_line(cmd, env=None, timeout=900):
    import json
    e = dict(os.environ)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    return json.loads(lines[-1])


def test_bench_through_the_multi_gpu_branch_on_rccl_at_world_1(hip_lib, cuda_device):
    """bench.py under the DRIVER's launcher form (`python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1`) with
    --force-distributed: the process group on the real `nccl` (= RCCL) backend, ShardedRows, the gather to rank 0, gather_check,
    the all_gather_object of the ranks' device identities and BASELINE configs[3] (`config4_strong`, 1,048,576 rows) all execute
    on the real engine -- every statement the 8-GPU run will execute, with N = 1.  (Scaling itself stays unmeasured until the
    driver's SCALE record holds N > 1.)"""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29641', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '2',
           '--force-distributed', '--cpu-seconds', '0']
    line = _bench_line(cmd)
    assert line['n_gpus'] == 1 and line['ranks_seen'] == 1 and line['scaling'] == 'weak'
    assert line['collectives']['backend'] == 'nccl' and line['collectives']['forced_at_world_1'] is True
    assert re.match(r'^\d+\.\d+', str(line['collectives']['rccl_version'])), line['collectives']
    assert 'uuid=' in line['ranks'][0]['device'] or 'pci=' in line['ranks'][0]['device'], line['ranks']
    assert line['gather_check']['ok'] is True and line['gather_check']['shards'] == 1 and line['gather_check']['rows'] == 65536
    assert line['gather_ms'] > 0
    c4 = line['config4_strong']
    assert c4['total_rows'] == 1048576 and c4['rows_per_gpu'] == 1048576 and c4['scaling'] == 'strong' and c4['value'] > 1e7
    assert line['parity']['max_abs_xyzds'] <= 1e-4           # the timed (gathered) step's own rows against the oracle
    assert 15e6 < line['value'] < 40e6, line['value']        # the gather of 1.3 MB on one device costs next to nothing


def test_bench_refuses_more_ranks_than_devices(hip_lib, cuda_device):
    """`--gpus N` on a node with fewer devices ends non-zero with ONE line naming both numbers, before anything is launched
    (self-launched form) or before the rendezvous (torchrun form: the rank without a device of its own)."""
    import torch
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 1), '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and 'device_count' in (r.stderr + r.stdout) and '--gpus %d' % (have + 1) in (r.stderr + r.stdout), r.stderr[-800:]
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    e = dict(os.environ, RANK=str(have), LOCAL_RANK=str(have), WORLD_SIZE=str(have + 1), LOCAL_WORLD_SIZE=str(have + 1),
             MASTER_ADDR='127.0.0.1', MASTER_PORT='29642')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 1), '--steps', '2', '--warmup', '1'],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=e)
    assert r.returncode != 0 and 'no HIP device of its own' in (r.stderr + r.stdout), r.stderr[-800:]
