"""-m gpu: the N > 1 code path on the REAL backend with the REAL engine, at world size 1 (one GPU per box):
`nccl` (= RCCL) process-group initialisation with device_id, the collective into slices of one pre-allocated tensor, and
the stream ordering between the engine's kernels (torch's current stream) and the collective.  Scaling itself is
unmeasured here: the driver's 8-GPU run is the only place a curve can come from."""
import os
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
