"""-m gpu: dense_mid_kernel (128 x 64 / 128 x 128 workgroup tiles for the batches whose 256x256 tiles are fewer than the CUs)
against fp64 and against the 256x256-tile kernel, layer by layer and through whole models on both sides of its row window."""
import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tile", ["mid64", "mid128", "half"])
@pytest.mark.parametrize("m,k,n", [(256, 64, 256), (700, 96, 256), (1000, 1024, 1024), (3000, 32, 512), (9000, 1024, 1024), (5000, 256, 512),
                                   (128, 192, 256), (4500, 2048, 512), (8192, 192, 1024)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
def test_mid_single_layer(hip_lib, cuda_device, tile, m, k, n, relu, res):
    """One layer: fp32-class accuracy against fp64, and the same operands through the 256x256-tile kernel (only the fp32
    summation order differs).  Row counts that are not multiples of the tile, K of one to 32 lines, in-place residual."""
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
    scale = max(1.0, np.abs(ref).max())
    y_mid = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile).cpu().numpy()
    y_tile = engine.debug_linear(xd, w, b, relu=relu, res=rd).cpu().numpy()
    assert np.abs(y_mid - ref).max() <= 4e-6 * scale
    assert np.abs(y_mid - y_tile).max() <= 2e-6 * scale


def test_mid_single_product_mode(hip_lib, cuda_device):
    from monoloco_amd import engine
    rng = np.random.default_rng(3)
    m, k, n = 2500, 512, 512
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    x = rng.standard_normal((m, k)).astype(np.float32)
    xd = torch.tensor(x).to(cuda_device)
    ref = np.maximum(x.astype(np.float64) @ w.astype(np.float64).T + b, 0)
    y_tile = engine.debug_linear(xd, w, b, relu=True, precision='f16').cpu().numpy()
    for tile in ('mid64', 'mid128'):
        y = engine.debug_linear(xd, w, b, relu=True, precision='f16', tile_kernel=tile).cpu().numpy()
        assert np.abs(y - ref).max() <= 2e-2
        assert np.abs(y - y_tile).max() <= 1e-4          # same rounded operands, one product per term
    with pytest.raises(Exception):
        engine.debug_linear(xd, w, b, relu=True, precision='bf16', tile_kernel='mid64')


@pytest.mark.parametrize("mode", ["mono", "stereo"])
@pytest.mark.parametrize("m", [513, 1000, 2049, 4096, 5000, 8192, 12288])
def test_mid_path_matches_tile_path(hip_lib, cuda_device, m, mode):
    """Whole model inside the window (small_rows, mid_rows]: the default route is dense_mid_kernel + separate head launches,
    mid_rows = 0 sends the same rows through the 256x256-tile kernels with their fused heads.  Each meets the parity bar
    against fp64, and they agree with each other far below it; both tile heights."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(4, in_features=in_f, out_features=out_f).items()}
    rng = np.random.default_rng(m)
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(sd, device=cuda_device)
    eng.set_tuning(mid_rows=0)
    raw_tile = eng.forward_raw(x).cpu()
    raws = {}
    for tile in (0, 64, 128, 256):   # (256: dense_kernel_w4's half-size tile for the long-K layers)
        eng.set_tuning(mid_rows=12288, mid_tile=tile)      # (the window's end is a tuning default)
        raws[tile] = eng.forward_raw(x).cpu()
    ref64 = O.loco_forward(sd, x.cpu(), dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    assert (raw_tile.double() - ref64).abs().max().item() <= 1e-4
    for tile, raw in raws.items():
        assert (raw.double() - ref64).abs().max().item() <= 1e-4, tile
        assert (raw - raw_tile).abs().max().item() <= 2e-6 * scale, tile
    # the automatic choice: dense_mid_kernel with 64-row tiles, 128-row tiles from one per CU, above 4096 rows the half-size w4 tile
    assert torch.equal(raws[0], raws[64 if m < 4096 else (128 if m == 4096 else 256)])
    eng.close()


def test_mid_path_pipeline_and_mc_dropout(hip_lib, cuda_device):
    """The fused mono pipeline (prep -> layers -> heads -> post) and the batched MC-dropout passes inside the window."""
    from monoloco_amd import engine
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(6).items()}
    kps = torch.tensor(synth.make_poses(5000, 3)).to(cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    eng = engine.LocoEngine(sd, device=cuda_device)           # 5000 rows: inside the default window (512, 8192]
    out_mid, xyzds_mid, raw_mid = [t.clone() for t in eng.forward_mono(kps, kinv, want_raw=True)]
    eng.set_tuning(mid_rows=0)
    out_tile, xyzds_tile, raw_tile = eng.forward_mono(kps, kinv, want_raw=True)
    assert (raw_mid - raw_tile).abs().max().item() <= 2e-6 * max(1.0, raw_tile.abs().max().item())
    assert (xyzds_mid - xyzds_tile).abs().max().item() <= 1e-4
    assert (out_mid.nan_to_num() - out_tile.nan_to_num()).abs().max().item() <= 1e-4
    # 20 stochastic passes over 256 persons = 5120 batched rows; the masks are counter-based (row, column, seed), so the two
    # routes drop the same units and the passes agree like the deterministic forward does
    _, p_tile = eng.epistemic_mono(kps[:256], kinv, 20, 0.2, want_passes=True)
    p_tile = p_tile.clone()
    eng.set_tuning(mid_rows=12288)
    _, p_mid = eng.epistemic_mono(kps[:256], kinv, 20, 0.2, want_passes=True)
    assert (p_mid - p_tile).abs().max().item() <= 4e-6 * max(1.0, p_tile.abs().max().item())
    assert p_mid.std(0).min().item() > 0          # the passes do differ from each other
    eng.close()


@pytest.mark.parametrize("m", [1, 16, 128, 129, 600, 3000])
def test_loco_forward_frame_entry_all_routes(hip_lib, cuda_device, m):
    """Loco.forward on Python lists (ml_loco_frame_mono: up to 128 persons without any copy operation -- the pinned keypoints are
    read and both result blocks written by kernels --, staged copies beyond) against the engine's device-tensor pipeline and
    ml_post_geometry on the same persons: the dictionary and the cached post_process geometry bit for bit, on the small-row
    route (<= 512), the mid-size route and both sides of the 128-person switch."""
    from monoloco_amd import engine
    from monoloco_amd.network import Loco
    from monoloco_amd.network.architectures import LocoModel
    from monoloco_amd.network.process import packed_to_dict
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(6).items()}
    model = LocoModel(34, 9, 1024)
    model.load_state_dict(sd)
    net = Loco(model=model, mode='mono', device=cuda_device)
    kps = synth.make_poses(m, 11)
    kk = synth.KITTI_K
    for rep in range(2):                                          # the second call reuses the cached staging buffers
        before = hip_lib.ml_debug_frames_without_copies()
        dic = net.forward(kps.tolist(), kk)
        assert hip_lib.ml_debug_frames_without_copies() - before == (1 if m <= 128 else 0)   # torch's pinned memory is recognised
        out, _, _ = net.engine.forward_mono(torch.tensor(kps).to(cuda_device), engine.inverse_intrinsics(kk))
        ref = packed_to_dict(out, 9)
        for k in ('h', 'w', 'l', 'ori', 'bi', 'xyzd', 'd'):
            assert torch.equal(dic[k], ref[k]), (m, rep, k)
        assert torch.equal(dic['yaw'][0], ref['yaw'][0]) and torch.equal(dic['yaw'][1], ref['yaw'][1])
        geo = engine.post_geometry(torch.tensor(kps), kk, d=out[:, 3].contiguous(), device=cuda_device).cpu()
        assert torch.equal(dic._geo[3], geo), (m, rep)


def test_mid_path_ten_output_model_through_the_mono_pipeline(hip_lib, cuda_device):
    """A 34 -> 10 LocoModel (w_fin with 9 outputs + the aux logit) through the fused mono pipeline inside the mid-size window:
    heads_pair_kernel<9> with the post-process riding in it, against the tile path."""
    from monoloco_amd import engine
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(7, in_features=34, out_features=10).items()}
    kps = torch.tensor(synth.make_poses(3000, 5)).to(cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    eng = engine.LocoEngine(sd, device=cuda_device)
    out_mid, xyzds_mid, raw_mid = [t.clone() for t in eng.forward_mono(kps, kinv, want_raw=True)]
    out_nr, _, _ = eng.forward_mono(kps, kinv)                     # without the raw rows: they never leave the registers
    assert torch.equal(out_nr.nan_to_num(), out_mid.nan_to_num())
    eng.set_tuning(mid_rows=0)
    out_tile, xyzds_tile, raw_tile = eng.forward_mono(kps, kinv, want_raw=True)
    assert raw_mid.shape == (3000, 10)
    assert (raw_mid - raw_tile).abs().max().item() <= 2e-6 * max(1.0, raw_tile.abs().max().item())
    assert (xyzds_mid - xyzds_tile).abs().max().item() <= 1e-4
    assert (out_mid.nan_to_num() - out_tile.nan_to_num()).abs().max().item() <= 1e-4
    eng.close()


@pytest.mark.parametrize("mode", ["mono", "stereo"])
@pytest.mark.parametrize("m", [513, 2049, 4096, 6000, 8192])
def test_mid_fused_heads_match_the_pair_kernel(hip_lib, cuda_device, m, mode):
    """Round 5: inside the mid window both heads ride in the dense epilogues (dense_mid_kernel<.., HEAD>, the half-size w4 tile with
    HEAD = -1 / 8 / 9) and tail_mono_kernel / the reduce kernels end the call -- against heads_pair_kernel behind the last layer
    (option mid_heads 0, rounds 3-4) and against fp64; the plan says which of the two ran."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(5, in_features=in_f, out_features=out_f).items()}
    rng = np.random.default_rng(m + 1)
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(sd, device=cuda_device)
    fam = 'half' if m > 4096 else ('mid128' if m > 2048 else 'mid64')
    assert eng.plan_for_rows(m, with_post=False).endswith("L6 %s+aux; L7 %s+fin%d; end=reduce" % (fam, fam, out_f - 1))
    raw_fused = eng.forward_raw(x).cpu()
    eng.set_option('mid_heads', 0)
    assert eng.plan_for_rows(m, with_post=False).endswith("L7 %s; end=heads_pair" % fam)
    raw_pair = eng.forward_raw(x).cpu()
    ref64 = O.loco_forward(sd, x.cpu(), dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    assert (raw_fused.double() - ref64).abs().max().item() <= 1e-4
    assert (raw_fused - raw_pair).abs().max().item() <= 2e-6 * scale
    if mode == "mono":   # the fused pipeline: the post-process rides in tail_mono_kernel / in the pair kernel
        kps = torch.tensor(synth.make_poses(m, 9)).to(cuda_device)
        kinv = engine.inverse_intrinsics(synth.KITTI_K)
        conf = torch.rand(m, device=cuda_device)
        out_p, xyz_p, raw_p = [t.clone() for t in eng.forward_mono(kps, kinv, box_conf=conf, want_raw=True)]
        eng.set_option('mid_heads', 1)
        assert eng.plan_for_rows(m).endswith("end=tail_mono")
        out_f_, xyz_f, raw_f = eng.forward_mono(kps, kinv, box_conf=conf, want_raw=True)
        assert (raw_f - raw_p).abs().max().item() <= 2e-6 * max(1.0, raw_p.abs().max().item())
        assert (xyz_f - xyz_p).abs().max().item() <= 2e-5
        assert (out_f_.nan_to_num() - out_p.nan_to_num()).abs().max().item() <= 1e-4
    eng.close()


@pytest.mark.parametrize("tile", ["mid64", "mid128"])
@pytest.mark.parametrize("m,k,n", [(256, 64, 256), (700, 96, 256), (1000, 1024, 1024), (2048, 1024, 1024), (4096, 1024, 1024), (3000, 32, 512),
                                   (513, 2048, 512), (128, 192, 256)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
def test_mid_loaders_agree(hip_lib, cuda_device, tile, m, k, n, relu, res):
    """Round 6: dense_mid_kernel's loader is LDS-DMA into a three-stage ring (the default); the rounds-3-5 loader (global -> VGPR ->
    ds_write, two stages) stays selectable.  Same operands, same fragment reads, same MFMA order: the SAME bits, for K of one line
    (a single k-step: the ring's requests past the end) up to 64, row counts that are no multiple of the tile, in-place residual."""
    from monoloco_amd import engine
    rng = np.random.default_rng(m + k + n + 7)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    x = (rng.standard_normal((m, k)) * 2).astype(np.float32)
    r = (rng.standard_normal((m, n)) * 3).astype(np.float32) if res else None
    xd = torch.tensor(x).to(cuda_device)
    rd = torch.tensor(r).to(cuda_device) if res else None
    y_dma = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile)
    y_reg = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile, mid_dma=False)
    assert torch.equal(y_dma, y_reg)
    for _ in range(3):
        assert torch.equal(engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile), y_dma)


@pytest.mark.parametrize("tile,ksplit,dma", [("mid64", 2, True), ("mid64", 4, True), ("mid128", 2, True), ("mid128", 4, True),
                                             ("mid64", 2, False), ("mid128", 4, False)])
@pytest.mark.parametrize("m,k,n", [(1000, 1024, 1024), (2048, 1024, 1024), (4096, 1024, 1024), (700, 256, 256), (3000, 512, 512), (513, 2048, 512)])
@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
def test_mid_split_k_single_layer(hip_lib, cuda_device, tile, ksplit, dma, m, k, n, relu, res):
    """dense_mid_kernel<.., SPLITK> (round 6): the reduction of every output tile cut into 2 / 4 k ranges, one workgroup each, the
    last arriver adds the partial tiles in split order and runs the epilogue.  Against fp64 at the fp32-class bar, <= 2e-6 (relative
    to the layer's largest output) from the unsplit kernel -- only the fp32 summation order differs -- and the SAME bits in every
    one of 6 repetitions (the sum order does not depend on which workgroup arrives last).  The layer: architectures.py:88-102."""
    from monoloco_amd import engine
    rng = np.random.default_rng(m + k + n + ksplit)
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
    scale = max(1.0, np.abs(ref).max())
    y_one = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile, mid_dma=dma).cpu().numpy()
    y_split = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile, ksplit=ksplit, mid_dma=dma).cpu().numpy()
    assert np.abs(y_split - ref).max() <= 4e-6 * scale
    assert np.abs(y_split - y_one).max() <= 2e-6 * scale
    if (k // 32) % ksplit == 0 and (k // 32) // ksplit >= 4:
        assert not np.array_equal(y_split, y_one) or k <= 64, "the split kernel did not run"
    for _ in range(5):
        again = engine.debug_linear(xd, w, b, relu=relu, res=rd, tile_kernel=tile, ksplit=ksplit, mid_dma=dma).cpu().numpy()
        assert np.array_equal(again, y_split)


@pytest.mark.parametrize("mode", ["mono", "stereo"])
@pytest.mark.parametrize("m", [513, 1024, 2048, 3000, 4096])
def test_mid_split_k_whole_model(hip_lib, cuda_device, m, mode):
    """Whole models through the mid window with the split reduction (auto, 2, 4 k ranges; both tile heights) against the unsplit
    route and the fp64 oracle, the fused heads riding in the last arriver's epilogue; 50 repetitions give the same bits (the arrival
    counters re-arm themselves, the sum order is fixed)."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    in_f, out_f = (34, 9) if mode == "mono" else (68, 10)
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(4, in_features=in_f, out_features=out_f).items()}
    rng = np.random.default_rng(m)
    x = torch.tensor((rng.standard_normal((m, in_f)) * 3).astype(np.float32), device=cuda_device)
    eng = engine.LocoEngine(sd, device=cuda_device)
    ref64 = O.loco_forward(sd, x.cpu(), dtype=torch.float64)
    scale = max(1.0, ref64.abs().max().item())
    for tile in (64, 128):
        eng.set_tuning(mid_tile=tile)
        eng.set_option('mid_splitk', 1)
        eng.set_option('mid_dma', 0)
        raw_reg = eng.forward_raw(x).cpu()
        eng.set_option('mid_dma', 1)
        raw_one = eng.forward_raw(x).cpu()
        assert torch.equal(raw_one, raw_reg), tile            # the two loaders: the same bits through a whole model
        assert (raw_one.double() - ref64).abs().max().item() <= 1e-4
        for sk in (-1, 2, 4):
            eng.set_option('mid_splitk', sk)
            raw = eng.forward_raw(x).cpu()
            assert (raw.double() - ref64).abs().max().item() <= 1e-4, (tile, sk)
            assert (raw - raw_one).abs().max().item() <= 4e-6 * scale, (tile, sk)
            for _ in range(50):
                assert torch.equal(eng.forward_raw(x).cpu(), raw), (tile, sk)
    eng.close()


@pytest.mark.parametrize("m", [257, 300, 1000, 2048, 2049, 3072, 4096, 5000, 8192])
def test_input_layer_preprocesses_its_own_persons(hip_lib, cuda_device, m):
    """Round 6 (an option, off by default: it measured no gain): inside the mid window the mono pipeline's input layer can compute
    preprocess_monoloco (reference process.py:47-67) for its own persons in its prologue (dense_mid_kernel<.., PREP>, option
    `mid_prep`) instead of reading prep_kernel's lines: the SAME bits in every
    output (packed rows, parity tensor, raw rows -- the box centres feed the back-projection) as with prep_kernel in front, row
    counts that are no multiple of the tile, both tile heights and the half-size-tile part of the window (whose K = 64 layer is
    dense_mid_kernel's too), and the oracle's bar on a sample."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(4).items()}
    eng = engine.LocoEngine(sd, device=cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    kps = torch.tensor(synth.make_poses(m, seed=m)).to(cuda_device)
    conf = torch.rand(m, device=cuda_device)
    got = {}
    for fused in (1, 0):
        eng.set_option('mid_prep', fused)
        out, xyzds, raw = eng.forward_mono(kps, kinv, box_conf=conf, want_raw=True)
        got[fused] = (out.clone(), xyzds.clone(), raw.clone())
    for a, b in zip(got[1], got[0]):
        assert torch.equal(a, b)
    idx = torch.arange(0, m, max(1, m // 200))
    ref = O.forward_mono(sd, kps[idx].cpu(), synth.KITTI_K, box_conf=conf[idx].cpu())
    assert (got[1][1][idx].cpu() - ref['xyzds']).abs().max().item() <= 1e-4
    assert (got[1][0][idx, 14:16].cpu() - O.get_keypoints(kps[idx].cpu(), 'center')).abs().max().item() == 0.0   # uc, vc: the box centres
    eng.close()
