"""-m gpu: every HIP kernel against the CPU oracle, through the C ABI."""
import numpy as np
import pytest
import torch

import synth
from oracle import monoloco_oracle as O

pytestmark = pytest.mark.gpu


def _sd_t(sd):
    return {k: torch.tensor(v) for k, v in sd.items()}


def test_dense_layout_asymmetric(hip_lib, cuda_device):
    """Transpose-detecting check of the MFMA fragment / C-D layouts: integer-valued asymmetric
    operands (exactly representable in fp16) must reproduce x.W^T + b exactly."""
    from monoloco_amd import engine
    m, k, n = 300, 96, 512
    x = (np.arange(m * k, dtype=np.float32).reshape(m, k) % 13) - 6 + (np.arange(m)[:, None] % 7)
    w = ((np.arange(n * k, dtype=np.float32).reshape(n, k) * 7) % 11) - 5 + (np.arange(n)[:, None] % 3)
    b = np.arange(n, dtype=np.float32) - 100
    y = engine.debug_linear(torch.tensor(x, device=cuda_device), w, b).cpu().numpy()
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    assert np.array_equal(y, ref.astype(np.float32))


@pytest.mark.parametrize("precision,tol", [("f16x2", 2e-6), ("f16", 3e-3)])
@pytest.mark.parametrize("m,k,n,relu,res", [(1000, 1024, 1024, True, True), (257, 34, 256, True, False),
                                             (4096, 1024, 1024, False, False), (64, 68, 512, False, True)])
def test_dense_vs_fp64(hip_lib, cuda_device, precision, tol, m, k, n, relu, res):
    from monoloco_amd import engine
    rng = np.random.default_rng(m + k)
    x = (rng.standard_normal((m, k)) * 1.5).astype(np.float32)
    w = rng.uniform(-1, 1, (n, k)).astype(np.float32) / np.sqrt(k)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32) if res else None
    y = engine.debug_linear(torch.tensor(x, device=cuda_device), w, b, relu=relu,
                            res=torch.tensor(r, device=cuda_device) if res else None, precision=precision)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    if relu:
        ref = np.maximum(ref, 0)
    if res:
        ref = ref + r
    err = np.abs(y.cpu().numpy() - ref).max()
    assert err < tol * max(1.0, np.abs(ref).max()), err


def test_dense_fp16_subnormal_operands(hip_lib, cuda_device):
    """The lo halves of small activations are fp16 subnormals; the MFMA must not flush them.
    x = 2^-3 + 3*2^-24 splits into hi = 2^-3, lo = 3*2^-24 (an exactly representable fp16 subnormal);
    with unit weights over k = 32 the exact answer 4 + 3*2^-19 is representable in fp32, and a
    flushed lo would give exactly 4."""
    from monoloco_amd import engine
    m, k, n = 256, 32, 256
    x = np.full((m, k), 2.0 ** -3 + 3 * 2.0 ** -24, dtype=np.float32)
    w = np.ones((n, k), dtype=np.float32)
    y = engine.debug_linear(torch.tensor(x, device=cuda_device), w, np.zeros(n, np.float32)).cpu().numpy()
    assert np.all(y == np.float32(4 + 3 * 2.0 ** -19)), (y.min(), y.max())


@pytest.mark.parametrize("m", [1, 16, 255, 256, 1000, 4097])
def test_preprocess_matches_oracle(hip_lib, cuda_device, m):
    from monoloco_amd import engine
    kps = synth.make_keypoints(m, seed=m)
    x, c = engine.preprocess_mono(torch.tensor(kps), synth.KITTI_K, device=cuda_device, want_centre=True)
    ref = O.preprocess_monoloco(torch.tensor(kps), synth.KITTI_K)
    cref = O.get_keypoints(torch.tensor(kps), 'center')
    assert torch.equal(c.cpu(), cref)
    d = (x.cpu() - ref).abs().max().item()
    assert d <= 1e-6, d  # at most one ulp at |x| < 16 (bit-exactness is checked on the goldens)


def test_stereo_pairs_matches_oracle(hip_lib, cuda_device):
    from monoloco_amd import engine
    kl, kr = synth.make_poses(13, 1), synth.make_poses(7, 2)
    xl = engine.preprocess_mono(torch.tensor(kl), synth.KITTI_K, device=cuda_device)
    xr = engine.preprocess_mono(torch.tensor(kr), synth.KITTI_K, device=cuda_device)
    rows = engine.stereo_pairs(xl, xr).cpu()
    ref, _ = O.preprocess_monstereo(torch.tensor(kl), torch.tensor(kr), synth.KITTI_K)
    assert (rows - ref).abs().max().item() <= 2e-6


def test_extract_outputs_matches_oracle(hip_lib, cuda_device):
    from monoloco_amd import engine
    m = 3000
    rng = np.random.default_rng(0)
    raw = rng.standard_normal((m, 10)).astype(np.float32)
    raw[:, 0] = rng.uniform(0.2, 2.9, m)          # theta
    raw[:, 1] = rng.uniform(1.2, 1.9, m)          # psi
    raw[:, 2] = rng.uniform(0.5, 60, m)           # d
    raw[:, 3] = rng.uniform(-5, 0.5, m)           # log b/d
    kps = synth.make_poses(m, 4)
    conf = rng.uniform(0.1, 1, m).astype(np.float32)
    centre = O.get_keypoints(torch.tensor(kps), 'center')
    out, xyzds = engine.extract_outputs_device(torch.tensor(raw, device=cuda_device), centre=centre,
                                               kk=synth.KITTI_K, box_conf=conf)
    out, xyzds = out.cpu(), xyzds.cpu()
    ref = O.extract_outputs(torch.tensor(raw))
    xyz, cf = O.back_project(torch.tensor(kps), synth.KITTI_K, ref['d'], ref['bi'], conf)
    from monoloco_amd._lib import OUT_COLS as C
    assert (out[:, C['d']] - ref['d'][:, 0]).abs().max() == 0
    assert (out[:, C['bi']] - ref['bi'][:, 0]).abs().max() <= 1e-5 * ref['bi'].abs().max()
    assert (out[:, 0:2] - ref['xyzd'][:, 0:2]).abs().max() <= 2e-5
    zref = ref['xyzd'][:, 2]
    ok = ~torch.isnan(zref)
    assert torch.equal(torch.isnan(out[:, 2]), ~ok) or (torch.isnan(out[:, 2]) != ~ok).sum() <= 2
    # z = sqrt(d^2-x^2-y^2): error amplified by d/z
    both = ok & ~torch.isnan(out[:, 2])
    amp = (ref['d'][:, 0] / zref.clamp_min(1e-3))[both]
    assert ((out[:, 2] - zref)[both].abs() / amp).max() <= 5e-5
    assert (out[:, C['yaw']] - ref['yaw'][0][:, 0]).abs().max() <= 2e-6
    assert (out[:, C['aux']] - ref['aux'][:, 0]).abs().max() <= 2e-6
    assert (out[:, 8:11] - torch.as_tensor(raw[:, 4:7])).abs().max() == 0
    assert (xyzds[:, 0:3] - xyz).abs().max() <= 1e-5
    assert ((out[:, C['conf']] - cf).abs() / cf.abs().clamp_min(1e-6)).max() <= 1e-5
    ego = out[:, C['yaw_ego']][both]
    eref = ref['yaw'][1][:, 0][both]
    dd = (ego - eref).abs()
    dd = torch.minimum(dd, (dd - 2 * np.pi).abs())
    assert (dd / amp.clamp_min(1)).max() <= 5e-5


@pytest.mark.parametrize("merge", [True, False])
@pytest.mark.parametrize("m", [16, 700])
def test_forward_mono_parity(hip_lib, cuda_device, merge, m):
    """Whole mono pipeline vs the fp32 oracle (= the reference's CPU path) and vs fp64 truth."""
    from monoloco_amd import engine
    sd = synth.make_state_dict(1)
    kps = synth.make_poses(m, 11)
    conf = np.linspace(0.2, 1, m).astype(np.float32)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device, merge_w2w3=merge)
    out, xyzds, raw = eng.forward_mono(torch.tensor(kps), engine.inverse_intrinsics(synth.KITTI_K), box_conf=conf,
                                       want_raw=True)
    out, xyzds, raw = out.cpu(), xyzds.cpu(), raw.cpu()
    ref = O.forward_mono(_sd_t(sd), torch.tensor(kps), synth.KITTI_K, box_conf=conf)
    ref64 = O.forward_mono(_sd_t(sd), torch.tensor(kps), synth.KITTI_K, box_conf=conf, dtype=torch.float64)
    e_raw = (raw - ref['raw']).abs().max().item()
    e_par = (xyzds - ref['xyzds']).abs().max().item()
    e_raw64 = (raw.double() - ref64['raw']).abs().max().item()
    noise = (ref['raw'].double() - ref64['raw']).abs().max().item()
    print("mono m=%d merge=%s: raw vs fp32 %.2e, vs fp64 %.2e (reference's own fp32 noise %.2e), xyzds %.2e"
          % (m, merge, e_raw, e_raw64, noise, e_par))
    assert e_par <= 1e-4      # the north-star bar on (x, y, z, d, sigma)
    assert e_raw <= 1e-4
    assert e_raw64 <= max(4 * noise, 2e-5)
    eng.close()


def test_forward_raw_matches_pipeline(hip_lib, cuda_device):
    from monoloco_amd import engine
    sd = synth.make_state_dict(2)
    kps = synth.make_poses(300, 5)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device)
    _, _, raw = eng.forward_mono(torch.tensor(kps), engine.inverse_intrinsics(synth.KITTI_K), want_raw=True)
    x = engine.preprocess_mono(torch.tensor(kps), synth.KITTI_K, device=cuda_device)
    raw2 = eng.forward_raw(x)
    assert (raw - raw2).abs().max().item() <= 1e-6
    eng.close()


def test_forward_stereo_parity(hip_lib, cuda_device):
    from monoloco_amd import engine
    sd = synth.make_state_dict(3, in_features=68, out_features=10)
    kl, kr = synth.make_poses(37, 1), synth.make_poses(9, 2)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device)
    res = eng.forward_stereo(torch.tensor(kl), torch.tensor(kr), engine.inverse_intrinsics(synth.KITTI_K),
                             want_raw_all=True)
    ref = O.forward_stereo(_sd_t(sd), torch.tensor(kl), torch.tensor(kr), synth.KITTI_K)
    raw_all = res['raw_all'].cpu()
    assert (raw_all - ref['raw_all']).abs().max().item() <= 1e-4
    assert int(res['ties'].item()) == 0
    best_ref = ref['mask'].float().argmax(1).int()
    # the arg-max may legitimately differ only where two logits are closer than the parity bar
    best = res['best'].cpu()
    diff = (best != best_ref).nonzero().flatten()
    aux = ref['raw_all'].view(37, 9, 10)[:, :, -1]
    for i in diff.tolist():
        assert abs(aux[i, best[i]] - aux[i, best_ref[i]]) < 1e-4
    same = best == best_ref
    assert (res['xyzds'].cpu()[same] - ref['xyzds'][same]).abs().max().item() <= 1e-4
    eng.close()


def test_empty_and_errors(hip_lib, cuda_device):
    from monoloco_amd import engine, _lib
    sd = synth.make_state_dict(1, hidden=256)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device)
    out, xyzds, _ = eng.forward_mono(torch.zeros((0, 3, 17)), engine.inverse_intrinsics(synth.KITTI_K))
    assert out.shape == (0, 16) and xyzds.shape == (0, 5)
    with pytest.raises(_lib.MonolocoHipError):
        eng.forward_stereo(torch.zeros((2, 3, 17)), torch.zeros((2, 3, 17)), engine.inverse_intrinsics(synth.KITTI_K))
    eng.close()
    bad = dict(sd)
    bad['w1.weight'] = np.zeros((250, 34), np.float32)
    with pytest.raises(_lib.MonolocoHipError):
        engine.LocoEngine(_sd_t(bad), device=cuda_device)


@pytest.mark.parametrize("mode", ["mono", "stereo"])
@pytest.mark.parametrize("m", [1, 16, 17, 100, 129, 300, 1000, 2048])
def test_small_row_path_matches_tile_path(hip_lib, cuda_device, monkeypatch, m, mode):
    """rows <= the small-row threshold (512 by default, 2048 here: LocoEngine.set_tuning) run dense_small_kernel (16x16 tiles up to 128 rows, 32x32 tiles above; K split over 4
    waves), larger batches the 256x256-tile persistent kernel: same operands and epilogue, only the fp32
    accumulation order differs.
    Both must agree with each other far below the parity bar and each must meet the bar against fp64."""
    from monoloco_amd import engine
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = synth.make_state_dict(4, in_features=in_f, out_features=out_f)
    rng = np.random.default_rng(m)
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device)
    eng.set_tuning(small_rows=0, mid_rows=0)
    raw_tile = eng.forward_raw(x).cpu()
    eng.set_tuning(small_rows=2048)
    raw_small = eng.forward_raw(x).cpu()
    ref64 = O.loco_forward(_sd_t(sd), x.cpu(), dtype=torch.float64)
    scale = ref64.abs().max().item()
    assert (raw_small.double() - ref64).abs().max().item() <= 1e-4
    assert (raw_tile.double() - ref64).abs().max().item() <= 1e-4
    assert (raw_small - raw_tile).abs().max().item() <= 2e-6 * max(1.0, scale)
    eng.close()


def test_row_chunking_is_bit_identical(hip_lib, cuda_device, monkeypatch):
    """Row chunking (LocoEngine.set_tuning(chunk_rows=...)) walks the batch in row chunks through all layers (Infinity-Cache residency experiment): rows
    are independent, so the result must not change by a bit -- including the fused-head partial sums."""
    from monoloco_amd import engine
    sd = synth.make_state_dict(6)
    kps = torch.tensor(synth.make_poses(5000, 3)).to(cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device)
    out0, xyzds0, raw0 = eng.forward_mono(kps, kinv, want_raw=True)
    out0, xyzds0, raw0 = out0.clone(), xyzds0.clone(), raw0.clone()
    eng.set_tuning(chunk_rows=2048)
    out1, xyzds1, raw1 = eng.forward_mono(kps, kinv, want_raw=True)
    eng.set_tuning(chunk_rows=0)
    assert torch.equal(raw0, raw1) and torch.equal(xyzds0, xyzds1)
    assert torch.equal(out0.nan_to_num(), out1.nan_to_num())
    eng.close()


@pytest.mark.parametrize("m", [16, 300])
def test_small_row_path_single_fp16_mode(hip_lib, cuda_device, monkeypatch, m):
    """ML_PREC_F16 (one MFMA per product, the comparison mode) through both small-row kernels: same single-product
    arithmetic as the tile kernel, so the two paths stay close to each other although both are ~1e-2 off fp64."""
    from monoloco_amd import engine
    sd = synth.make_state_dict(8)
    rng = np.random.default_rng(m)
    x = torch.tensor((rng.standard_normal((m, 34)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(_sd_t(sd), device=cuda_device, precision='f16')
    eng.set_tuning(small_rows=0, mid_rows=0)
    raw_tile = eng.forward_raw(x).cpu()
    eng.set_tuning(small_rows=2048)
    raw_small = eng.forward_raw(x).cpu()
    ref64 = O.loco_forward(_sd_t(sd), x.cpu(), dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    assert (raw_small.double() - ref64).abs().max().item() <= 5e-2 * scale
    assert (raw_small - raw_tile).abs().max().item() <= 2e-3 * scale
    eng.close()


def test_handles_do_not_leak_device_memory(hip_lib, cuda_device):
    """ml_loco_create / finalize / reserve / forward / destroy in a loop (and the trainer likewise): the free device
    memory must come back."""
    from monoloco_amd import engine
    from monoloco_amd.train import HipTrainer
    sd = _sd_t(synth.make_state_dict(9))
    kps = torch.tensor(synth.make_poses(3000, 2)).to(cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    x = torch.randn(400, 34, device=cuda_device)
    y = torch.randn(400, 11, device=cuda_device).abs() + 0.5
    xb = torch.randn(4200, 34, device=cuda_device)
    yb = torch.randn(4200, 11, device=cuda_device).abs() + 0.5

    def cycle():
        eng = engine.LocoEngine(sd, device=cuda_device, reserve_rows=4096)
        eng.forward_mono(kps, kinv)
        eng.epistemic_mono(kps[:64], kinv, 5)
        eng.close()
        tr = HipTrainer(sd, p_dropout=0.2, lr=0.001, device=cuda_device, auto_tune_mtl=True)
        tr.step(x, y)
        tr.step(xb, yb)     # >= 4096 rows: the line / transposed-line / packed-weight buffers of the 3-product route as well
        tr.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(cuda_device)[0]
    for _ in range(12):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(cuda_device)[0]
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)


def test_post_geometry_block_matches_oracle_and_strided_distances(hip_lib, cuda_device):
    """ml_post_geometry(_strided): the per-person geometry of Loco.post_process (reference net.py:195-215) against the
    oracle's get_keypoints / pixel_to_camera / xyz_from_distance; distances read from a column of a packed (m, 16) block
    (what Loco.forward does) give the same bits as a contiguous vector."""
    from monoloco_amd import engine
    m = 300
    kps = torch.tensor(synth.make_keypoints(m, seed=9))
    d = torch.rand(m) * 40 + 0.5
    geo = engine.post_geometry(kps.to(cuda_device), synth.KITTI_K, d.to(cuda_device), device=cuda_device).cpu()
    packed = torch.zeros((m, 16), device=cuda_device)
    packed[:, 3] = d.to(cuda_device)
    out = torch.empty((m, 12), device=cuda_device)
    geo2 = engine.post_geometry(kps.to(cuda_device), synth.KITTI_K, packed[:, 3], device=cuda_device, out=out)
    assert geo2 is out and torch.equal(geo2.cpu(), geo)
    for col, mode in ((0, 'shoulder'), (2, 'head'), (4, 'center')):
        ref = O.get_keypoints(kps, mode)
        assert (geo[:, col:col + 2] - ref).abs().max() <= 3e-4, mode      # pixels up to ~1200: 1-2 ulp of a 5-joint mean
    xy = O.pixel_to_camera(O.get_keypoints(kps, 'center'), synth.KITTI_K, 1)
    assert (geo[:, 6:9] - xy).abs().max() <= 1e-6
    xyz = O.xyz_from_distance(d.view(-1, 1), xy)
    assert (geo[:, 9:12] - xyz).abs().max() <= 1e-4 * 40
    assert engine.post_geometry(kps[:0].to(cuda_device), synth.KITTI_K, None, device=cuda_device).shape == (0, 12)


@pytest.mark.parametrize("mode", ["mono", "stereo"])
@pytest.mark.parametrize("m", [65, 80, 100, 128])
def test_small_multi_row_tiles_same_bits(hip_lib, cuda_device, m, mode):
    """Round 5: above 64 rows the small-row layers run dense_small_multi_kernel (a workgroup keeps its 16 weight rows in registers
    and walks several row tiles) -- same operands, same per-tile arithmetic and summation order as one 16 x 16 tile per workgroup:
    the raw outputs are bit-identical to that route, and within the usual bar of the 32 x 32-tile route and of fp64."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(8, in_features=in_f, out_features=out_f).items()}
    rng = np.random.default_rng(m)
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(sd, device=cuda_device)
    assert "small16" in eng.plan_for_rows(m, with_post=False)
    raw_multi = eng.forward_raw(x).cpu()
    eng.set_option('small_multi', 0)
    eng.set_tuning(small32_rows=100000)
    raw_t16 = eng.forward_raw(x).cpu()
    eng.set_tuning(small32_rows=0)
    raw_t32 = eng.forward_raw(x).cpu()
    assert torch.equal(raw_multi, raw_t16)
    ref64 = O.loco_forward(sd, x.cpu(), dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    assert (raw_multi.double() - ref64).abs().max().item() <= 1e-4
    assert (raw_multi - raw_t32).abs().max().item() <= 2e-6 * scale
    eng.close()
