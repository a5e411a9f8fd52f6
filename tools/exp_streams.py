"""Experiment: does de-synchronising the workgroups' epilogue bursts pay?  N engines on N streams, each a 1/N row chunk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import torch, synth
from monoloco_amd import engine
dev = torch.device('cuda', 0)
_t0 = time.perf_counter(); torch.cuda._sleep(10_000_000); torch.cuda.synchronize(); print('calibration: _sleep(1e7) = %.2f ms' % ((time.perf_counter() - _t0) * 1e3))
sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}
kinv = engine.inverse_intrinsics(synth.KITTI_K)
M = 65536
kps = torch.tensor(synth.make_keypoints(M, seed=1)).to(dev)
TILE_KERNEL = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for n in (1, 2, 4):
    m = M // n
    engs = [engine.LocoEngine(sd, device=dev, reserve_rows=m) for _ in range(n)]
    for e in engs:
        e.set_tuning(tile_kernel=TILE_KERNEL)
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    outs = [(torch.empty((m, 16), device=dev), torch.empty((m, 5), device=dev)) for _ in range(n)]
    chunks = [kps[i * m:(i + 1) * m].contiguous() for i in range(n)]
    def step():
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                engs[i].forward_mono(chunks[i], kinv, out=outs[i][0], xyzds=outs[i][1])
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    off_us = float(os.environ.get('STREAM_OFFSET_US', '0'))
    if off_us > 0:   # enforce a phase offset between the streams (persists: every stream runs back to back from here on)
        for i in range(n):
            with torch.cuda.stream(streams[i]):
                torch.cuda._sleep(int(i * off_us * 1e-6 * 100e6))   # _sleep counts 100 MHz ticks on ROCm (calibrated below)
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("streams %d x %6d rows: %.3f ms per 65536 rows  (%.2f M persons/s)" % (n, m, dt * 1e3, M / dt / 1e6), flush=True)
    for e in engs:
        e.close()
