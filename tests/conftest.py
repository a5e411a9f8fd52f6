import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _host_threads():
    """Cores this process may use, capped by the container's cgroup CPU quota (the GPU boxes show 256 cores and grant 16: torch's
    default of one thread per visible core makes the CPU oracle several times slower there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(_host_threads())


@pytest.fixture(scope='session')
def hip_lib():
    """The C-ABI library; built on demand (hipcc cross-compiles without a GPU)."""
    from monoloco_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


@pytest.fixture(scope='session')
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return torch.device('cuda', 0)
