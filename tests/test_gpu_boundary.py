"""-m gpu: the boundary pieces of round 4 -- Loco.epistemic_uncertainty(inputs), filter_outputs' mask on the device,
the Trainer's host statistics as one launch, the batch collation kernel, chunked validation."""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _model(mode):
    from monoloco_amd.network.architectures import LocoModel
    sd = {k: torch.tensor(v) for k, v in np.load(os.path.join(G, 'ckpt_%s_h256.npz' % mode)).items()}
    m = LocoModel(68 if mode == 'stereo' else 34, 10 if mode == 'stereo' else 9, sd['w1.weight'].shape[0])
    m.load_state_dict(sd)
    return m


def test_epistemic_uncertainty_method_takes_preprocessed_inputs(hip_lib, cuda_device):
    """reference net.py:135-161: the public method takes the (m, 34) network inputs; forward() calls it on
    preprocess_monoloco's result (net.py:126-128) -- same passes, masks and draws, so the same numbers."""
    from monoloco_amd.network import Loco
    from monoloco_amd.network.process import preprocess_monoloco
    gold = np.load(os.path.join(G, 'golden_path.npz'))
    kps = torch.tensor(gold['mono_kps'][:200])
    net = Loco(model=_model('mono'), mode='mono', device=cuda_device, n_dropout=12, p_dropout=0.2)
    dic = net.forward(kps, synth.KITTI_K)
    inputs = preprocess_monoloco(kps.to(cuda_device), synth.KITTI_K)
    assert inputs.shape == (200, 34)
    varss = net.epistemic_uncertainty(inputs)
    assert varss.shape == (200,) and varss.device == inputs.device      # on self.device, like the reference's
    assert torch.equal(varss.cpu(), dic['epi'])
    assert torch.equal(net.epistemic_uncertainty(inputs.cpu()).cpu(), dic['epi'])   # host inputs are accepted as well
    # more passes than fit one batched launch (131072 rows / 200 persons = 655 passes per launch)
    net.n_dropout = 700
    big = net.epistemic_uncertainty(inputs)
    assert torch.isfinite(big).all() and (big > 0).all()
    ster = Loco(model=_model('stereo'), mode='stereo', device=cuda_device, n_dropout=3)
    with pytest.raises(AssertionError):
        ster.epistemic_uncertainty(torch.zeros(4, 68))


def test_stereo_tied_rows_is_filter_outputs_mask(hip_lib, cuda_device):
    """process.py:319-327 on crafted logits: clear winners, exact ties (2 and 3 rows), a NaN candidate, all equal."""
    from monoloco_amd import engine
    rng = np.random.default_rng(5)
    for ml, mr in ((7, 4), (300, 5), (1, 1), (513, 3)):
        raw = rng.standard_normal((ml, mr, 10)).astype(np.float32)
        if ml >= 7:
            raw[1, :, -1] = 0.25                              # all tied
            raw[2, 0, -1] = raw[2, mr - 1, -1] = 9.0          # two tied (first and last)
            raw[3, 1, -1] = np.nan                            # NaN: the reference keeps nothing for this person
            raw[5, :, -1] = [3.0, 3.0, 3.0, -1.0, 2.0][:mr]   # three (or mr) tied
            raw[ml - 1, mr - 1, -1] = 50.0
        t = torch.tensor(raw)
        val = t[:, :, -1]
        mask = val >= val.max(dim=1, keepdim=True).values
        want = mask.reshape(-1).nonzero().flatten().int()
        got = engine.stereo_tied_rows(t.reshape(ml * mr, 10).to(cuda_device), ml, mr)
        assert got.dtype == torch.int32 and torch.equal(got.cpu(), want), (ml, mr)


def _ref_stats(out, lab, stereo):
    """The reference Trainer's host arithmetic, restated with torch (trainer.py:213-232, losses.py:85-96,112-131)."""
    r = {'d_val': (out[:, 2:3] - lab[:, 3:4]).abs().mean().item()}
    for t, c in (('x', 0), ('y', 1), ('h', 4), ('w', 5), ('l', 6)):
        r[t] = (out[:, c] - lab[:, c]).abs().mean().item()
    r['ori_val'] = (torch.atan2(out[:, 7], out[:, 8]) - torch.atan2(lab[:, 7], lab[:, 8])).abs().mean().item() * 180 / 3.14
    norm = 1 - out[:, 2:3] / lab[:, 3:4]
    r['d'] = (norm.abs() * torch.exp(-out[:, 3:4]) + 0.01 + out[:, 3:4] + 2).mean().item()
    r['ori'] = (out[:, 7:9] - lab[:, 7:9]).abs().mean().item()
    errs = (out[:, 2:3] - lab[:, 3:4]).abs()
    bis = torch.exp(out[:, 3:4]) * out[:, 2:3]
    r['bi'] = bis.mean().item()
    r['bi%'] = float((errs <= bis).sum()) / errs.shape[0]
    r['std'] = errs.std().item()
    if stereo:
        r['aux'] = torch.nn.functional.binary_cross_entropy_with_logits(out[:, 9:10], lab[:, 10:11]).item()
        mask = (torch.sigmoid(out[:, 9:10]) >= 0.5).float()
        r['aux_acc'] = 1. - (mask - lab[:, 10:11]).abs().mean().item()
    return r


@pytest.mark.parametrize("m,stereo", [(169, False), (1, True), (5000, True), (100000, False)])
def test_val_stats_matches_the_trainers_torch_arithmetic(hip_lib, cuda_device, m, stereo):
    from monoloco_amd import engine
    g = torch.Generator().manual_seed(m)
    C, L = (10, 11) if stereo else (9, 10)
    out = torch.randn(m, C, generator=g)
    out[:, 2] = out[:, 2].abs() * 10 + 1
    out[:, 3] = out[:, 3] * 0.3 - 1
    lab = torch.randn(m, L, generator=g)
    lab[:, 3] = lab[:, 3].abs() * 10 + 1
    if stereo:
        lab[:, 10] = (torch.rand(m, generator=g) > 0.5).float()
    got = engine.val_stats(out.to(cuda_device), lab.to(cuda_device))
    ref = _ref_stats(out.double(), lab.double(), stereo)
    for k, v in ref.items():
        if k == 'std' and m == 1:
            assert np.isnan(got['std']) and np.isnan(v)
            continue
        assert abs(got[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, got[k], v)
    if not stereo:
        assert got['aux'] == 0.0 and got['aux_acc'] == 0.0


def test_gather_rows_is_index_select(hip_lib, cuda_device):
    from monoloco_amd import engine
    g = torch.Generator().manual_seed(3)
    for width in (34, 11, 68):
        src = torch.randn(700, width, generator=g)
        idx = torch.randperm(700, generator=g)[:331]
        got = engine.gather_rows(src.to(cuda_device), idx)
        assert torch.equal(got.cpu(), src.index_select(0, idx))
    assert engine.gather_rows(src.to(cuda_device), idx[:0]).shape == (0, 68)


def test_trainer_eval_walks_a_large_set_in_chunks(hip_lib, cuda_device):
    """Trainer.evaluate() hands the whole validation set to ml_trainer_eval in one call: chunked inside (8192 rows),
    the means equal the one-launch statistics of the very outputs it returned."""
    from monoloco_amd import engine
    from monoloco_amd.network.architectures import LocoModel
    from monoloco_amd.train.hip_trainer import HipTrainer
    torch.manual_seed(2)
    model = LocoModel(34, 9, 256)
    tr = HipTrainer(model.state_dict(), p_dropout=0.2, lr=1e-3, sched_gamma=0.9, sched_step=20, seed=1, device=cuda_device)
    assert tr.can_evaluate
    m = 20000
    g = torch.Generator().manual_seed(9)
    x = torch.randn(m, 34, generator=g).to(cuda_device)
    y = torch.randn(m, 10, generator=g)
    y[:, 3] = y[:, 3].abs() * 10 + 1
    y = y.to(cuda_device)
    plain, raw = tr.evaluate_batch(x, y, want_outputs=True)
    ref = engine.val_stats(raw, y)
    for k in ('d', 'x', 'y', 'h', 'w', 'l', 'ori', 'd_val', 'ori_val'):
        assert abs(plain[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), (k, plain[k], ref[k])
    # chunk boundaries do not show in the outputs: the same rows alone give the same bits
    _, raw_a = tr.evaluate_batch(x[8192:8192 + 300], y[8192:8192 + 300], want_outputs=True)
    assert torch.equal(raw_a, raw[8192:8192 + 300])
    tr.close()
    # a width the trainer's own evaluation does not take: the library says so itself
    tr2 = HipTrainer(LocoModel(34, 9, 200).state_dict(), p_dropout=0.2, lr=1e-3, sched_gamma=0.9, sched_step=20, seed=1,
                     device=cuda_device)
    assert not tr2.can_evaluate
    tr2.close()


def test_route_query_follows_the_tuning(hip_lib, cuda_device):
    """ml_loco_route: the dense kernel family a forward of `rows` rows takes (bench.py labels its lines with it)."""
    from monoloco_amd import engine
    sd = {k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 1024).items()}
    eng = engine.LocoEngine(sd, device=cuda_device)
    got = [eng.route_for_rows(r) for r in (16, 128, 129, 256, 257, 512, 513, 2048, 2304, 3072, 4096, 4097, 6144, 8192, 8193, 65536)]
    # (round 6, hidden 1024: the small-row kernels hand over at 256 rows, the 128-row tiles start where the 64-row tiles would need a second
    #  wave on the CUs = above 2048 rows)
    assert got == ['small16', 'small16', 'small32', 'small32', 'mid64', 'mid64', 'mid64', 'mid64', 'mid128', 'mid128', 'mid128', 'half', 'half', 'half',
                   'tile', 'tile']
    narrow = engine.LocoEngine({k: torch.tensor(v) for k, v in synth.make_state_dict(1, 34, 9, 256).items()}, device=cuda_device)
    assert narrow.route_for_rows(512) == 'small32' and narrow.route_for_rows(513) == 'mid64'   # narrower models keep the 512-row hand-over
    narrow.close()
    eng.set_tuning(mid_tile=64)
    assert eng.route_for_rows(6144) == 'mid64'
    eng.set_tuning(mid_tile=256)
    assert eng.route_for_rows(1024) == 'half'
    eng.set_tuning(mid_rows=0)
    assert eng.route_for_rows(1024) == 'tile' and eng.route_for_rows(100) == 'small16'
    eng.close()
    bf = engine.LocoEngine(sd, device=cuda_device, precision='bf16')
    assert bf.route_for_rows(100) == 'tile'          # the bf16 comparison mode exists on the tile path only
    bf.close()


def test_frame_completion_word(hip_lib, cuda_device):
    """Round 5: a single image's forward tells the host it is done through a word of pinned memory (the last launch releases it behind
    every store of the frame; ml_loco_frame_mono polls it) instead of hipStreamSynchronize.  2000 frames of changing content and
    person counts: every dictionary equals the one the stream-synchronised route returns, no frame times out."""
    import copy
    import json
    import os
    from monoloco_amd.network import Loco, load_calibration, preprocess_pifpaf
    from monoloco_amd.network.architectures import LocoModel
    model = LocoModel(34, 9, 1024)
    model.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_dict(2).items()})
    net = Loco(model=model, mode='mono', device=cuda_device)
    kk = synth.KITTI_K
    t0 = hip_lib.ml_debug_frame_spin(1)
    rng = np.random.default_rng(0)
    frames = [synth.make_poses(int(rng.integers(1, 40)), seed=100 + i).tolist() for i in range(40)]
    ref = []
    hip_lib.ml_debug_frame_spin(0)                     # the reference: stream synchronisation
    for kps in frames:
        ref.append({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in net.forward(kps, kk).items()})
    hip_lib.ml_debug_frame_spin(1)
    for rep in range(50):
        for kps, want in zip(frames, ref):
            got = net.forward(kps, kk)
            for key in ('xyzd', 'd', 'bi', 'h', 'w', 'l', 'ori'):
                assert torch.equal(got[key], want[key]), (rep, key)
            assert torch.equal(got['yaw'][0], want['yaw'][0]) and torch.equal(got['yaw'][1], want['yaw'][1])
    # (a frame falls back to the stream synchronisation when its word has not arrived after 5 ms -- e.g. the polling thread was
    #  descheduled that long; a broken mechanism would time out on all 2000)
    assert hip_lib.ml_debug_frame_spin(-1) - t0 <= 5


def test_stereo_frame_entry_matches_the_general_route(hip_lib, cuda_device):
    """Round 5: MonStereo's Loco.forward on host keypoints goes through ml_loco_frame_stereo (one call, pinned buffers, completion
    word) -- same dictionary and geometry block as the general route (device tensors in: ml_loco_forward_stereo +
    ml_post_geometry_strided + copies), for lists / numpy, with and without right keypoints; a frame with tied aux logits falls back
    and keeps every tied pair row like the reference (process.py:325-326)."""
    from monoloco_amd.network import Loco
    from monoloco_amd.network.architectures import LocoModel
    model = LocoModel(68, 10, 1024)
    model.load_state_dict({k: torch.tensor(v) for k, v in synth.make_state_dict(3, 68, 10, 1024).items()})
    net = Loco(model=model, mode='stereo', device=cuda_device)
    kk = synth.KITTI_K
    t0 = hip_lib.ml_debug_frame_spin(-1)
    for ml, mr, seed in ((16, 5, 1), (1, 1, 2), (30, 30, 3), (7, 0, 4), (3, 40, 5)):
        kl = synth.make_poses(ml, seed)
        kr = synth.make_poses(mr, seed + 50) if mr else None
        if kr is not None:
            kr[:, 0] -= 15.0
        want = net.forward(torch.tensor(kl).to(cuda_device), kk, keypoints_r=None if kr is None else torch.tensor(kr).to(cuda_device))
        for as_list in (True, False):
            got = net.forward(kl.tolist() if as_list else kl, kk, keypoints_r=None if kr is None else (kr.tolist() if as_list else kr))
            assert list(got.keys()) == list(want.keys())
            for key in ('h', 'w', 'l', 'ori', 'aux', 'bi', 'xyzd', 'd'):
                assert got[key].shape == want[key].shape and torch.equal(got[key], want[key]), (ml, mr, key)
            assert torch.equal(got['yaw'][0], want['yaw'][0]) and torch.equal(got['yaw'][1], want['yaw'][1])
            assert got['epi'] == want['epi'] and torch.equal(got._geo[3], want._geo[3])
    assert hip_lib.ml_debug_frame_spin(-1) - t0 <= 1
    # two identical right poses: every left person's best aux logit is tied -> both pair rows are kept (2 ml rows)
    kl = synth.make_poses(6, 9)
    kr = np.repeat(synth.make_poses(1, 10), 2, axis=0)
    tied = net.forward(kl.tolist(), kk, keypoints_r=kr.tolist())
    assert tied['d'].shape[0] == 12
    want = net.forward(torch.tensor(kl).to(cuda_device), kk, keypoints_r=torch.tensor(kr).to(cuda_device))
    assert torch.equal(tied['xyzd'], want['xyzd'])
