"""-m gpu: dense_kernel_w4 (4 waves x 128x128 wave tiles, one wave per SIMD, 4-slot hi/lo LDS ring, continuous k-stream)
against fp64 and against dense_kernel_pp, layer by layer and through whole models, incl. several tiles per workgroup."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


KERNELS = {4: 'w4', 2: 'pp'}   # debug_linear's tile_kernel argument; engines select per handle (LocoEngine.set_tuning)


@pytest.mark.parametrize("m,k,n", [(256, 64, 256), (700, 96, 256), (1000, 1024, 1024), (3000, 256, 512), (70000, 128, 1024)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
def test_w4_single_layer(hip_lib, cuda_device, m, k, n, relu, res):
    """One layer: fp32-class accuracy against fp64, and the same operands through dense_kernel_pp (only the fp32
    summation order differs).  70000 x 1024 outputs = 1096 tiles: every workgroup walks 4-5 tiles of the stream."""
    from monoloco_amd import engine
    rng = np.random.default_rng(m + k + n)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    x = (rng.standard_normal((m, k)) * 2).astype(np.float32)
    r = (rng.standard_normal((m, n)) * 3).astype(np.float32) if res else None
    xd = torch.tensor(x).to(cuda_device)
    rd = torch.tensor(r).to(cuda_device) if res else None
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if relu:
        ref = np.maximum(ref, 0)
    if res:
        ref = ref + r
    out = {}
    for kern in (4, 2):
        out[kern] = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=KERNELS[kern]).cpu().numpy()
        err = np.abs(out[kern] - ref).max()
        assert err <= 4e-6 * max(1.0, np.abs(ref).max()), (kern, err)
    assert np.abs(out[4] - out[2]).max() <= 2e-6 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_w4_single_product_modes(hip_lib, cuda_device, precision):
    from monoloco_amd import engine
    rng = np.random.default_rng(3)
    m, k, n = 2500, 512, 512
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    x = rng.standard_normal((m, k)).astype(np.float32)
    xd = torch.tensor(x).to(cuda_device)
    out = {}
    for kern in (4, 2):
        out[kern] = engine.debug_linear(xd, w, b, relu=True, precision=precision, tile_kernel=KERNELS[kern]).cpu().numpy()
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0)
    tol = 2e-2 if precision == 'f16' else 1.5e-1
    assert np.abs(out[4] - ref).max() <= tol
    # same rounded operands, same single product per term: only the summation order (and, for bf16, a final-ulp tie)
    assert np.abs(out[4] - out[2]).max() <= (1e-4 if precision == 'f16' else 2 ** -6 * np.abs(ref).max())


@pytest.mark.parametrize("mode", ["mono", "stereo"])
def test_w4_whole_model_vs_pp_and_oracle(hip_lib, cuda_device, mode):
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(4, in_features=in_f, out_features=out_f).items()}
    rng = np.random.default_rng(11)
    m = 20000   # 79 row panels x 4 column tiles = 316 tiles > 256 workgroups: the stream crosses tile boundaries
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    raw = {}
    for merge in (True, False):
        eng = engine.LocoEngine(sd, device=cuda_device, merge_w2w3=merge)
        for kern in (4, 2):
            eng.set_tuning(tile_kernel=kern, everywhere=True)   # incl. the input layer and the fused-head layer
            raw[(merge, kern)] = eng.forward_raw(x).cpu()
        eng.set_tuning(tile_kernel=4)                           # the default mix of the two kernels
        raw[(merge, 'mix')] = eng.forward_raw(x).cpu()
        eng.close()
    idx = torch.arange(0, m, 41)
    ref64 = O.loco_forward(sd, x.cpu()[idx], dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    for key, r in raw.items():
        assert (r[idx].double() - ref64).abs().max().item() <= 1e-4, key
    assert (raw[(True, 4)] - raw[(True, 2)]).abs().max().item() <= 4e-6 * scale
    assert (raw[(False, 4)] - raw[(False, 2)]).abs().max().item() <= 4e-6 * scale


def test_w4_is_deterministic_and_row_independent(hip_lib, cuda_device):
    """Same bits run after run and under a row permutation (rows are independent; a race between the DMA ring and the
    fragment reads would show up as run-to-run differences in some tile)."""
    from monoloco_amd import engine
    eng = engine.LocoEngine({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}, device=cuda_device)
    eng.set_tuning(tile_kernel=4, everywhere=True)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    m = 65536
    kps = torch.tensor(synth.make_keypoints(m, seed=21)).to(cuda_device)
    _, x0, r0 = eng.forward_mono(kps, kinv, want_raw=True)
    x0, r0 = x0.clone(), r0.clone()
    for _ in range(5):
        _, x1, r1 = eng.forward_mono(kps, kinv, want_raw=True)
        assert torch.equal(r1, r0) and torch.equal(x1, x0)
    perm = torch.randperm(m, device=cuda_device)
    _, x2, r2 = eng.forward_mono(kps[perm].contiguous(), kinv, want_raw=True)
    assert torch.equal(r2, r0[perm])
    eng.close()


@pytest.mark.parametrize("rows", [65536, 8192])
def test_w4_repeat_run_stress_on_two_streams(hip_lib, cuda_device, rows):
    """The repeat-run detector that found the xgemm prologue race (profiles/r03_xgemm_occupancy.md), pointed at dense_kernel_w4's
    hand-counted LDS-DMA ring: 200 forwards -- two engines on two HIP streams, launched alternately so that their workgroups
    share CUs, L2 and the memory system in ever-changing phase -- must all produce the bits of the first, quiet run.  A wait that
    is one DMA group short, a hazard pad that a new compiler no longer respects or a slot refilled one barrier early shows up
    here as a handful of differing 32 x 32 sub-tiles in some run; the test reports how many rows differed and in which run.
    65536 rows: the full-size tile walking several tiles per workgroup; 8192 rows: the half-size tile (NJ = 2)."""
    from monoloco_amd import engine
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    engs = [engine.LocoEngine(sd, device=cuda_device, reserve_rows=rows) for _ in range(2)]
    for e in engs:
        if rows > 8192:
            e.set_tuning(tile_kernel=4, everywhere=True)
    kps = [torch.tensor(synth.make_keypoints(rows, seed=31 + i)).to(cuda_device) for i in range(2)]
    want = []
    for e, k in zip(engs, kps):
        _, _, r = e.forward_mono(k, kinv, want_raw=True)
        want.append(r.clone())
    torch.cuda.synchronize(cuda_device)
    assert engs[0].route_for_rows(rows) == ('tile' if rows > 8192 else 'half')
    streams = [torch.cuda.Stream(cuda_device) for _ in range(2)]
    outs = [[torch.empty((rows, 16), dtype=torch.float32, device=cuda_device) for _ in range(4)] for _ in range(2)]
    xyz = [[torch.empty((rows, 5), dtype=torch.float32, device=cuda_device) for _ in range(4)] for _ in range(2)]
    raws = [[torch.empty_like(want[0]) for _ in range(4)] for _ in range(2)]
    bad = []
    n_runs = 100                      # per stream: 200 forwards = 1400 w4 launches at 65536 rows
    for base in range(0, n_runs, 4):
        for j in range(4):
            for s in range(2):        # alternate the two streams launch by launch
                with torch.cuda.stream(streams[s]):
                    engs[s].forward_mono(kps[s], kinv, out=outs[s][j], xyzds=xyz[s][j], raw=raws[s][j])
        torch.cuda.synchronize(cuda_device)
        for j in range(4):
            for s in range(2):
                if not torch.equal(raws[s][j], want[s]):
                    bad.append((base + j, s, int((raws[s][j] != want[s]).any(1).sum())))
    for e in engs:
        e.close()
    assert not bad, "runs whose bits differ (run, stream, rows differing): %s" % bad[:10]
