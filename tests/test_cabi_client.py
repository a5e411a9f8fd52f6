"""The C ABI used from plain C (tests/cabi/client.c): no Python or torch objects cross the boundary.  The client is
compiled with gcc against include/monoloco_hip.h, fed a state_dict dump and a test case, and its device result is
checked (inside the client) against the CPU oracle's (x, y, z, d, sigma)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

import synth
from oracle import monoloco_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'cabi', 'client.c')
LIBDIR = os.path.join(ROOT, 'monoloco_amd', 'lib')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')


def _build(tmp_path):
    exe = str(tmp_path / 'cabi_client')
    cmd = ['gcc', '-O1', '-std=c11', '-Wall', '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROCM, 'include'), SRC,
           '-o', exe, '-L' + LIBDIR, '-lmonoloco_hip', '-L' + os.path.join(ROCM, 'lib'), '-lamdhip64', '-lm',
           '-Wl,-rpath,' + LIBDIR, '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_c_client_compiles_against_the_header(hip_lib, tmp_path):
    """CPU: the header is valid C11 and the library exports what the client links against."""
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_client_runs_the_mono_pipeline(hip_lib, cuda_device, tmp_path):
    exe = _build(tmp_path)
    m, hidden = 1500, 256
    sd = synth.make_state_dict(5, 34, 9, hidden)
    with open(tmp_path / 'weights.bin', 'wb') as f:
        keys = [k for k in sd if not k.endswith('num_batches_tracked')]
        f.write(struct.pack('<i', len(keys)))
        for k in keys:
            a = np.ascontiguousarray(sd[k], dtype=np.float32)
            f.write(struct.pack('<i', len(k)) + k.encode() + struct.pack('<q', a.size) + a.tobytes())
    kps = synth.make_poses(m, 11).astype(np.float32)
    conf = np.linspace(0.2, 1.0, m).astype(np.float32)
    ref = O.forward_mono({k: torch.tensor(v) for k, v in sd.items()}, torch.tensor(kps), synth.KITTI_K, box_conf=conf)
    kinv = np.linalg.inv(np.asarray(synth.KITTI_K, dtype=np.float64))
    from monoloco_amd import engine
    kinv = engine.inverse_intrinsics(synth.KITTI_K)   # the fp32 inverse the reference computes (camera.py:23)
    with open(tmp_path / 'case.bin', 'wb') as f:
        f.write(struct.pack('<4iq', 34, hidden, 9, 3, m))
        f.write(kps.tobytes() + np.asarray(kinv, dtype=np.float32).tobytes() + conf.tobytes())
        f.write(ref['xyzds'].numpy().astype(np.float32).tobytes())
    # a fresh process without torch: the library binds to /opt/rocm's HIP runtime here
    res = subprocess.run([exe, str(tmp_path / 'weights.bin'), str(tmp_path / 'case.bin'), '1e-4'], capture_output=True,
                         text=True, timeout=300)
    sys.stdout.write(res.stdout + res.stderr)
    assert res.returncode == 0, res.stdout + res.stderr
