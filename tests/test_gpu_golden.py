"""-m gpu: the HIP path (through the C ABI and the reference-shaped Python surface) against the golden
vectors recorded from the real reference, plus size-independent properties at the full batch size."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4  # the north-star bar: abs deviation on (x, y, z, d, sigma) and on the raw outputs


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(os.path.join(G, 'golden_path.npz')))


@pytest.fixture(scope='module')
def c1():
    return json.load(open(os.path.join(G, 'golden_c1.json'))), dict(np.load(os.path.join(G, 'golden_c1.npz')))


def _weights(tag, mode):
    if tag == 'A':
        sd = synth.make_state_dict(1, 34, 9, 1024) if mode == 'mono' else synth.make_state_dict(3, 68, 10, 1024)
    else:
        sd = dict(np.load(os.path.join(G, 'ckpt_%s_h256.npz' % mode)))
    return {k: torch.tensor(v) for k, v in sd.items()}


def _model(tag, mode):
    from monoloco_amd.network.architectures import LocoModel
    sd = _weights(tag, mode)
    m = LocoModel(68 if mode == 'stereo' else 34, 10 if mode == 'stereo' else 9, sd['w1.weight'].shape[0])
    m.load_state_dict(sd)
    return m


def test_preprocess_bit_exact_on_fixture_rows(hip_lib, cuda_device, gold):
    """Data pin of the reference's fixtures: X == preprocess_monoloco(kps, K), bit for bit, from the HIP kernel."""
    from monoloco_amd.network.process import preprocess_monoloco, preprocess_monstereo
    kps = torch.tensor(gold['mono_kps'])
    for i, k in enumerate(gold['mono_unique_k']):
        rows = gold['mono_k_index'] == i
        if rows.any():
            x = preprocess_monoloco(kps[rows], k.tolist())
            assert x.device.type == 'cpu'
            assert torch.equal(x, torch.tensor(gold['mono_x_fixture'][rows]))
    x = preprocess_monoloco(kps.to(cuda_device), synth.KITTI_K)
    assert x.is_cuda and torch.equal(x.cpu(), torch.tensor(gold['mono_x_kitti']))
    nl, nr = gold['stereo_ava_nl_nr']
    rows, clusters = preprocess_monstereo(torch.tensor(gold['stereo_kps_l'][:nl]), torch.tensor(gold['stereo_kps_r'][:nr]),
                                          synth.KITTI_K)
    assert clusters == [int(nr)] * int(nl)
    assert torch.equal(rows, torch.tensor(gold['stereo_ava_inputs']))


def test_reference_unit_test_pixel_to_camera(hip_lib, cuda_device):
    """reference tests/test_utils.py:18-25, run against the HIP-backed pixel_to_camera."""
    from monoloco_amd.utils import pixel_to_camera
    uv = [1000., 400.]
    a = pixel_to_camera(uv, synth.KITTI_K, 1)[0] * 10
    b = pixel_to_camera(uv, synth.KITTI_K, 10)[0]
    assert torch.equal(a, b)


def test_utils_against_oracle(hip_lib, cuda_device, gold):
    from monoloco_amd import utils as U
    from oracle import monoloco_oracle as O
    kps = torch.tensor(gold['mono_kps'])
    for mode in ('center', 'bottom', 'head', 'shoulder', 'hip', 'ankle'):
        assert (U.get_keypoints(kps, mode) - O.get_keypoints(kps, mode)).abs().max() <= 2.5e-4  # 2 ulp at 1238 px (mean order)
    assert torch.equal(U.get_keypoints(kps, 'center'), O.get_keypoints(kps, 'center'))
    assert U.get_keypoints(kps[0], 'center').shape == (1, 2)
    uv = O.get_keypoints(kps, 'center')
    assert torch.equal(U.pixel_to_camera(uv, synth.KITTI_K, 1), O.pixel_to_camera(uv, synth.KITTI_K, 1))
    assert torch.equal(U.pixel_to_camera(kps[:, 0:2, :], synth.KITTI_K, 10), O.pixel_to_camera(kps[:, 0:2, :], synth.KITTI_K, 10))
    xy = O.pixel_to_camera(uv, synth.KITTI_K, 1)
    d = torch.linspace(1, 40, len(kps))
    assert (U.xyz_from_distance(d, xy) - O.xyz_from_distance(d, xy)).abs().max() <= 4e-6
    assert (U.xyz_from_distance(7.5, xy[3]) - O.xyz_from_distance(7.5, xy[3])).abs().max() <= 1e-6
    rtp = torch.stack((torch.linspace(0.2, 2.9, 50), torch.linspace(1.2, 1.9, 50), torch.linspace(1, 50, 50)), 1)
    x_ref = rtp[:, 2] * torch.sin(rtp[:, 1]) * torch.cos(rtp[:, 0])
    assert (U.to_cartesian(rtp, 'x')[:, 0] - x_ref).abs().max() <= 1e-5
    yaw = torch.linspace(-3, 3, 50).view(-1, 1)
    xyz = torch.stack((torch.linspace(-5, 5, 50), torch.ones(50), torch.linspace(1, 30, 50)), 1)
    ego = yaw + torch.atan2(xyz[:, 0], xyz[:, 2]).view(-1, 1)
    ego = torch.where(ego > np.pi, ego - 2 * np.pi, ego)
    ego = torch.where(ego < -np.pi, ego + 2 * np.pi, ego)
    assert (U.back_correct_angles(yaw, xyz) - ego).abs().max() <= 1e-6


@pytest.mark.parametrize("tag", ["A", "B"])
def test_fixture_poses_vs_reference(hip_lib, cuda_device, gold, tag):
    """500 real KITTI poses of the reference's fixture: raw outputs and the parity tensor vs the reference."""
    from monoloco_amd import engine
    p = 'mono_%s_' % tag
    eng = engine.LocoEngine(_weights(tag, 'mono'), device=cuda_device)
    out, xyzds, raw = eng.forward_mono(torch.tensor(gold['mono_kps']), engine.inverse_intrinsics(synth.KITTI_K),
                                       box_conf=gold['mono_conf'], want_raw=True)
    out, xyzds, raw = out.cpu().numpy(), xyzds.cpu().numpy(), raw.cpu().numpy()
    e_raw = np.abs(raw - gold[p + 'raw']).max()
    e_raw64 = np.abs(raw - gold[p + 'raw64']).max()
    noise = np.abs(gold[p + 'raw'] - gold[p + 'raw64']).max()
    ref_par = np.concatenate((gold[p + 'xyz_pred'], gold[p + 'd'], gold[p + 'bi']), 1)
    e_par = np.abs(xyzds - ref_par).max()
    print("fixture poses W-%s: raw vs ref fp32 %.2e, vs ref fp64 %.2e (ref fp32-vs-fp64 %.2e), xyzds %.2e"
          % (tag, e_raw, e_raw64, noise, e_par))
    assert e_raw <= TOL and e_par <= TOL
    assert e_raw64 <= max(3 * noise, 2e-5)  # not worse than the reference's own rounding noise class
    assert np.abs(out[:, 0:2] - gold[p + 'xyzd'][:, 0:2]).max() <= TOL
    assert np.abs(out[:, 8:11] - np.concatenate((gold[p + 'h'], gold[p + 'w'], gold[p + 'l']), 1)).max() <= TOL
    assert np.abs(out[:, 5] - gold[p + 'yaw_pred'][:, 0]).max() <= TOL
    rel = np.abs(out[:, 11] / gold[p + 'conf'] - 1)
    assert rel.max() <= 1e-4
    # spherical z = sqrt(d^2-x^2-y^2) is ill-conditioned (amplification d/z): report against conditioning
    z_ref, d_ref = gold[p + 'xyzd'][:, 2], gold[p + 'xyzd'][:, 3]
    both = ~np.isnan(z_ref) & ~np.isnan(out[:, 2])
    amp = (np.abs(d_ref) / np.maximum(z_ref, 1e-3))[both]
    assert (np.abs(out[:, 2] - z_ref)[both] / np.maximum(amp, 1)).max() <= TOL
    eng.close()


@pytest.mark.parametrize("tag", ["A", "B"])
def test_loco_dropin_on_pifpaf_fixture(hip_lib, cuda_device, c1, tag):
    """BASELINE config 1: the reference's call sequence (predict.py:226-236) on tests/002282.png.pifpaf.json."""
    from monoloco_amd.network import Loco, load_calibration, preprocess_pifpaf
    cj, cn = c1
    ann = json.load(open(os.path.join(G, 'pifpaf_002282.json')))
    boxes, kps = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
    kk = load_calibration('kitti', (1238, 374))
    net = Loco(model=_model(tag, 'mono'), mode='mono', device=cuda_device)
    dic = net.forward(kps, kk)
    assert list(dic.keys()) == ['h', 'w', 'l', 'ori', 'bi', 'xyzd', 'd', 'yaw', 'epi']
    for key in ('h', 'w', 'l', 'ori', 'bi', 'd'):
        t = dic[key]
        assert t.device.type == 'cpu' and t.dtype == torch.float32 and tuple(t.shape) == cn['fwd_%s_%s' % (tag, key)].shape
        assert np.abs(t.numpy() - cn['fwd_%s_%s' % (tag, key)]).max() <= TOL, key
    assert np.abs(dic['xyzd'].numpy()[:, [0, 1, 3]] - cn['fwd_%s_xyzd' % tag][:, [0, 1, 3]]).max() <= TOL
    assert np.abs(dic['yaw'][0].numpy() - cn['fwd_%s_yaw_pred' % tag]).max() <= TOL
    assert dic['epi'] == [0.] * 16 and isinstance(dic['epi'], list)
    pp = net.post_process(dic, boxes, kps, kk)
    ref = cj['post_%s' % tag]
    assert set(pp.keys()) == set(ref.keys())
    assert pp['aux'] == [] and pp['gt'] == ref['gt'] and pp['boxes'] == ref['boxes'] and pp['uv_kps'] == ref['uv_kps']
    for key in ('uv_centers', 'uv_shoulders', 'uv_heads'):
        assert pp[key] == ref[key], key
    for key in ('dds_pred', 'stds_ale', 'xyz_pred', 'angles'):
        assert np.abs(np.array(pp[key]) - np.array(ref[key])).max() <= TOL, key
    assert np.abs(np.array(pp['confs']) / np.array(ref['confs']) - 1).max() <= 1e-4
    assert pp['stds_epi'] == ref['stds_epi']
    # forward() hands post_process the geometry block of these keypoints (one device round trip less per frame); a plain
    # dict with the same entries, or other keypoint objects, take the stand-alone route: identical results
    assert getattr(dic, '_geo', None) is not None and isinstance(dic, dict)
    assert dict(net.post_process(dict(dic), boxes, kps, kk)) == dict(pp)
    assert dict(net.post_process(dic, boxes, copy.deepcopy(kps), kk)) == dict(pp)
    assert json.loads(json.dumps({k: v for k, v in dic.items() if k == 'epi'})) == {'epi': [0.] * 16}
    # with ground truth (IoU matching, re-ordering, xyz_real)
    ppg = net.post_process(dic, boxes, kps, kk, dic_gt=cj['dic_gt'])
    refg = cj['post_gt_%s' % tag]
    assert ppg['gt'] == refg['gt'] and ppg['boxes'] == refg['boxes'] and ppg['boxes_gt'] == refg['boxes_gt']
    assert ppg['dds_real'] == refg['dds_real']
    assert np.abs(np.array(ppg['xyz_real']) - np.array(refg['xyz_real'])).max() <= 1e-5
    assert np.abs(np.array(ppg['xyz_pred']) - np.array(refg['xyz_pred'])).max() <= TOL
    # empty input and the no-prediction branch
    assert net.forward([], kk) is None
    assert dict(net.post_process(None, [], [], kk)) == {}


@pytest.mark.parametrize("tag", ["A", "B"])
def test_stereo_vs_reference(hip_lib, cuda_device, gold, tag):
    from monoloco_amd import engine
    from monoloco_amd.network import Loco
    p = 'stereo_%s_' % tag
    eng = engine.LocoEngine(_weights(tag, 'stereo'), device=cuda_device)
    raw = eng.forward_raw(torch.tensor(gold['stereo_x_fixture'])).cpu().numpy()
    e = np.abs(raw - gold[p + 'raw_fixture']).max()
    noise = np.abs(gold[p + 'raw_fixture'] - gold[p + 'raw64_fixture']).max()
    print("stereo fixture rows W-%s: raw vs ref fp32 %.2e, vs fp64 %.2e (ref noise %.2e)"
          % (tag, e, np.abs(raw - gold[p + 'raw64_fixture']).max(), noise))
    assert e <= TOL
    eng.close()
    nl, nr = (int(v) for v in gold['stereo_ava_nl_nr'])
    net = Loco(model=_model(tag, 'stereo'), mode='stereo', device=cuda_device)
    dic = net.forward(gold['stereo_kps_l'][:nl].tolist(), synth.KITTI_K, keypoints_r=gold['stereo_kps_r'][:nr].tolist())
    assert list(dic.keys()) == ['h', 'w', 'l', 'ori', 'aux', 'bi', 'xyzd', 'd', 'yaw', 'epi']
    # pairs are selected by arg-max of the aux logit; a different pick is legitimate only within TOL of a tie
    ref_all = gold[p + 'ava_raw_all'].reshape(nl, nr, 10)
    gap = np.sort(ref_all[:, :, -1], 1)
    clear = (gap[:, -1] - gap[:, -2]) > 2 * TOL
    assert clear.sum() >= nl // 2
    for key in ('d', 'bi', 'h', 'w', 'l', 'aux'):
        assert np.abs(dic[key].numpy() - gold[p + 'ava_' + key])[clear].max() <= TOL, key
    if tag == 'A':
        dic0 = net.forward(gold['stereo_kps_l'][:5].tolist(), synth.KITTI_K)
        assert np.abs(dic0['d'].numpy() - gold['stereo_A_noright_d']).max() <= TOL
        assert len(dic0['epi']) == 5


def test_stereo_exact_ties_keep_all_rows(hip_lib, cuda_device, gold):
    """Two identical right poses tie exactly: the reference returns both pair rows per left person."""
    from monoloco_amd.network import Loco
    net = Loco(model=_model('B', 'stereo'), mode='stereo', device=cuda_device)
    kl = gold['stereo_kps_l'][:6]
    kr = np.stack((gold['stereo_kps_r'][0], gold['stereo_kps_r'][0]))
    dic = net.forward(kl.tolist(), synth.KITTI_K, keypoints_r=kr.tolist())
    assert dic['d'].shape[0] == 12
    assert torch.equal(dic['d'][0::2], dic['d'][1::2])


def test_full_batch_properties(hip_lib, cuda_device):
    """BASELINE config 2 size (65536 persons): rows are independent, so the result must be bit-identical
    under any row permutation and under batching, and every row must equal the same row computed alone
    in a smaller batch of the same kernel (and, within rounding, in the small-row kernel)."""
    from monoloco_amd import engine
    m = 65536
    eng = engine.LocoEngine({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}, device=cuda_device,
                            reserve_rows=m)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    kps = torch.tensor(synth.make_keypoints(m, seed=9)).to(cuda_device)
    conf = torch.rand(m, device=cuda_device)
    out, xyzds, raw = eng.forward_mono(kps, kinv, box_conf=conf, want_raw=True)
    out, xyzds, raw = out.clone(), xyzds.clone(), raw.clone()
    assert torch.isfinite(raw).all() and torch.isfinite(xyzds).all()
    perm = torch.randperm(m, device=cuda_device)
    out_p, xyzds_p, raw_p = eng.forward_mono(kps[perm].contiguous(), kinv, box_conf=conf[perm].contiguous(), want_raw=True)
    assert torch.equal(raw_p, raw[perm]) and torch.equal(xyzds_p, xyzds[perm])
    assert torch.equal(out_p.nan_to_num(), out[perm].nan_to_num())
    sub = slice(12345, 12345 + 9777)   # beyond the mid-size window (8192 rows): same tile kernels, so the same bits
    out_s, xyzds_s, raw_s = eng.forward_mono(kps[sub].contiguous(), kinv, box_conf=conf[sub].contiguous(), want_raw=True)
    assert torch.equal(raw_s, raw[sub]) and torch.equal(xyzds_s, xyzds[sub])
    # a video batch takes dense_mid_kernel, a single image's worth of rows dense_small_kernel: same arithmetic, another fp32
    # summation order
    for lo, n in ((20000, 2777), (4321, 777), (4321, 300)):
        sub = slice(lo, lo + n)
        out_s, xyzds_s, raw_s = eng.forward_mono(kps[sub].contiguous(), kinv, box_conf=conf[sub].contiguous(), want_raw=True)
        assert (raw_s - raw[sub]).abs().max().item() <= 2e-6 * max(1.0, raw.abs().max().item()), n
        assert (xyzds_s - xyzds[sub]).abs().max().item() <= 5e-5, n  # a few fp32 ulps at 20-60 m
    # idempotence: same input, same bits
    out2, xyzds2, _ = eng.forward_mono(kps, kinv, box_conf=conf)
    assert torch.equal(xyzds2, xyzds)
    # a CPU-checkable sample of the big batch against the oracle
    from oracle import monoloco_oracle as O
    idx = torch.arange(0, m, 97)[:600]
    ref = O.forward_mono({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}, kps[idx].cpu(),
                         synth.KITTI_K, box_conf=conf[idx].cpu())
    assert (xyzds[idx].cpu() - ref['xyzds']).abs().max().item() <= TOL
    assert (raw[idx].cpu() - ref['raw']).abs().max().item() <= TOL
    eng.close()


def test_mc_dropout_epistemic(hip_lib, cuda_device, gold):
    """Loco(n_dropout > 0): MC-dropout spread of the distance (reference net.py:135-161).  The RNG is not
    torch's, so parity is statistical.  The reference draws its 100 Laplace samples with the same seed on
    every pass, so each person's value is dominated by the luck of those 100 draws (0.3x..1.8x the ideal
    sigma with torch's stream); what can be compared is (a) the statistics of the stochastic passes
    themselves against the oracle's torch-dropout passes, (b) the estimator against its closed form
    Var(mu) + 2 E[b^2] evaluated on the same passes, (c) the p -> 0 limit, (d) reproducibility."""
    from monoloco_amd import engine
    from monoloco_amd.network import Loco
    from oracle import monoloco_oracle as O
    sd = _weights('B', 'mono')
    kps = torch.tensor(gold['mono_kps'][:256])
    net = Loco(model=_model('B', 'mono'), mode='mono', device=cuda_device, n_dropout=20, p_dropout=0.2)
    dic = net.forward(kps, synth.KITTI_K)
    epi = dic['epi']
    assert isinstance(epi, torch.Tensor) and epi.shape == (256,) and epi.device.type == 'cpu'
    assert torch.isfinite(epi).all() and (epi > 0).all()
    eng = engine.LocoEngine(sd, device=cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    n_pass = 300
    epi, passes = eng.epistemic_mono(kps, kinv, n_pass, 0.2, want_passes=True)
    epi, passes = epi.cpu(), passes.cpu()
    x = O.preprocess_monoloco(kps, synth.KITTI_K)
    torch.manual_seed(0)
    ref_passes = torch.stack([O.loco_forward_mc(sd, x, 0.2) for _ in range(n_pass)])
    for col in (2, 3):  # d and log-b: same mean and spread over the passes as torch's dropout gives
        a, b = passes[:, :, col], ref_passes[:, :, col]
        r = a.std(0) / b.std(0)
        z = (a.mean(0) - b.mean(0)).abs() / (b.std(0) / n_pass ** 0.5)
        assert abs(r.median().item() - 1) < 0.03 and r.min() > 0.7 and r.max() < 1.4, (r.median(), r.min(), r.max())
        assert z.max() < 6.0, z.max()
    mu, bb = passes[:, :, 2], (torch.exp(passes[:, :, 3]) * passes[:, :, 2]).abs()
    ideal = torch.sqrt(mu.var(0) + 2 * (bb ** 2).mean(0))
    rr = epi / ideal
    print("MC-dropout: HIP estimate / closed form on its passes: median %.3f range %.2f..%.2f" % (rr.median(), rr.min(), rr.max()))
    assert abs(rr.median().item() - 1) < 0.06 and rr.min() > 0.5 and rr.max() < 1.7
    # p -> 0: every pass identical: epi = |bi| * std(fixed standard Laplace draws) ~ sqrt(2) |bi|
    bi = dic['bi'][:, 0]
    e0 = eng.epistemic_mono(kps, kinv, 3, p_dropout=1e-7).cpu()
    r0 = e0 / bi.abs()
    assert abs(r0.mean().item() - 2 ** 0.5) < 0.06 and r0.min() > 0.8 and r0.max() < 2.4
    assert torch.equal(e0, eng.epistemic_mono(kps, kinv, 3, p_dropout=1e-7).cpu())  # counter-based RNG
    eng.close()


def test_legacy_monoloco_model_vs_reference(hip_lib, cuda_device):
    """Legacy MonolocoModel (architectures.py:105-176) on the HIP engine: module forward, and Loco(net='monoloco')
    (zero-centred inputs, d / bi dictionary, net.py:95-100) against the reference's own classes."""
    from monoloco_amd.network import Loco
    from monoloco_amd.network.architectures import MonolocoModel
    g = dict(np.load(os.path.join(G, 'golden_legacy.npz')))
    model = MonolocoModel(input_size=34, output_size=2, linear_size=256)
    model.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd.')})
    model.eval()
    raw = model(torch.tensor(g['x']).to(cuda_device)).cpu().numpy()
    noise = np.abs(g['raw'] - g['raw64']).max()
    print("legacy MonolocoModel: raw vs ref fp32 %.2e, vs ref fp64 %.2e (ref noise %.2e)"
          % (np.abs(raw - g['raw']).max(), np.abs(raw - g['raw64']).max(), noise))
    assert np.abs(raw - g['raw']).max() <= TOL
    assert np.abs(raw - g['raw64']).max() <= max(3 * noise, 2e-5)
    net = Loco(model=model, mode='mono', net='monoloco', device=cuda_device)
    dic = net.forward(g['kps'].tolist(), g['kk'].tolist())
    assert set(dic) == {'d', 'bi', 'epi'} and dic['epi'] == [0.] * len(g['kps'])
    assert dic['d'].shape == g['d'].shape and not dic['d'].is_cuda
    assert np.abs(dic['d'].numpy() - g['d']).max() <= TOL
    assert np.abs(dic['bi'].numpy() - g['bi']).max() <= TOL
    # the fused pp pipeline must refuse a legacy model loudly
    from monoloco_amd import _lib, engine
    with pytest.raises(_lib.MonolocoHipError):
        net.engine.forward_mono(torch.tensor(g['kps']), engine.inverse_intrinsics(g['kk'].tolist()))
    # MC-dropout on the legacy net: columns (d, s) = 0:2, dropout only after the first layer
    net_mc = Loco(model=model, mode='mono', net='monoloco', device=cuda_device, n_dropout=30)
    epi = net_mc.forward(g['kps'].tolist(), g['kk'].tolist())['epi']
    assert epi.shape == (len(g['kps']),) and torch.isfinite(epi).all()
    lap = np.sqrt(2.0) * np.abs(g['bi'][:, 0])           # std of Laplace(d, bi) without any dropout spread
    assert (epi.numpy() > 0.5 * lap).all() and (epi.numpy() < 5.0 * lap + 5.0).all()


def test_legacy_monoloco_p_vs_reference(hip_lib, cuda_device):
    """extract_outputs_mono on the HIP kernel (bit-level against the reference on the same raw rows) and
    Loco(net='monoloco_p') = MonolocoModel(34 -> 256 -> 9) + extract_outputs_mono on the pifpaf fixture."""
    from monoloco_amd.network import Loco
    from monoloco_amd.network.architectures import MonolocoModel
    from monoloco_amd.network.process import extract_outputs_mono
    g = dict(np.load(os.path.join(G, 'golden_mono_p.npz')))
    dic = extract_outputs_mono(torch.tensor(g['raw']))
    for key in ('xyz', 'zb', 'h', 'w', 'l', 'ori'):
        assert torch.equal(dic[key], torch.tensor(g['ex_' + key])), key
    assert not dic['d'].is_cuda and dic['xyzd'].shape == (len(g['raw']), 4)
    assert np.abs(dic['d'].numpy() - g['ex_d']).max() <= 4e-6            # fp32 norm: summation order / sqrt ulp
    assert np.abs(dic['xyzd'].numpy() - g['ex_xyzd']).max() <= 4e-6
    assert np.abs(dic['bi'].numpy() / g['ex_bi'] - 1).max() <= 2e-6      # exp: libm vs device ulp
    assert np.abs(dic['yaw'][0].numpy() - g['ex_yaw_pred']).max() <= 1e-6
    assert np.abs(dic['yaw'][1].numpy() - g['ex_yaw_ego']).max() <= 2e-6
    slices = extract_outputs_mono(torch.tensor(g['raw']), tasks=('xyz', 'ori'))
    assert torch.equal(slices[0], torch.tensor(g['raw'][:, 0:3])) and torch.equal(slices[1], torch.tensor(g['raw'][:, 7:9]))

    model = MonolocoModel(input_size=34, output_size=9, linear_size=256)
    model.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd.')})
    net = Loco(model=model, mode='mono', net='monoloco_p', device=cuda_device)
    out = net.forward(g['kps'].tolist(), g['kk'].tolist())
    assert set(out) == {'xyz', 'zb', 'h', 'w', 'l', 'ori', 'xyzd', 'd', 'bi', 'yaw', 'epi'}
    for key in ('xyz', 'zb', 'h', 'w', 'l', 'ori', 'xyzd', 'd', 'bi'):
        assert np.abs(out[key].numpy() - g['net_' + key]).max() <= TOL, key
    assert np.abs(out['yaw'][0].numpy() - g['net_yaw_pred']).max() <= TOL
    assert np.abs(out['yaw'][1].numpy() - g['net_yaw_ego']).max() <= TOL


def test_laplace_sampling_statistics(hip_lib, cuda_device):
    """laplace_sampling (process.py:101-122) on the device generator: layout (n_samples, m), same draws on every
    call (the reference re-seeds with 1), Laplace(mu, |b|) moments per person."""
    from monoloco_amd.network.process import laplace_sampling
    m, n = 64, 20000
    mu = torch.linspace(1., 40., m)
    b = torch.linspace(0.05, 3., m) * torch.where(torch.arange(m) % 2 == 0, 1., -1.)   # sign is ignored (abs)
    outputs = torch.stack((mu, b), 1)
    xx = laplace_sampling(outputs, n)
    assert xx.shape == (n, m) and not xx.is_cuda
    assert torch.equal(xx, laplace_sampling(outputs, n))
    assert laplace_sampling(outputs.to(cuda_device), 8).is_cuda
    mean, std = xx.double().mean(0), xx.double().std(0)
    se = b.abs().double() * np.sqrt(2.0 / n)
    assert ((mean - mu.double()).abs() <= 5 * se).all()
    assert (std / (np.sqrt(2.0) * b.abs().double()) - 1).abs().max().item() <= 0.05
    med = xx.double().median(0).values
    assert ((med - mu.double()).abs() <= 5 * b.abs().double() / np.sqrt(n)).all()
    # two persons never share their draws
    z = (xx - mu) / b.abs()
    assert abs(np.corrcoef(z[:, 0].numpy(), z[:, 1].numpy())[0, 1]) < 0.03


def test_dataset_prep_rows_one_launch(hip_lib, cuda_device, gold):
    """Batched dataset preparation: all 500 mono / 556 stereo fixture rows, each with the K of its own image, in
    ONE launch -- bit for bit the X the reference's prep stored in its joints files."""
    from monoloco_amd.network.process import preprocess_monoloco_rows
    x = preprocess_monoloco_rows(torch.tensor(gold['mono_kps']), [k.tolist() for k in gold['mono_unique_k']],
                                 gold['mono_k_index'])
    assert x.shape == (500, 34) and x.device.type == 'cpu'
    assert torch.equal(x, torch.tensor(gold['mono_x_fixture']))
    xs = preprocess_monoloco_rows(torch.tensor(gold['stereo_kps_l']).to(cuda_device),
                                  [k.tolist() for k in gold['stereo_unique_k']], gold['stereo_k_index'],
                                  keypoints_r=torch.tensor(gold['stereo_kps_r']))
    assert xs.shape == (556, 68) and xs.is_cuda
    assert torch.equal(xs.cpu(), torch.tensor(gold['stereo_x_fixture']))
    assert preprocess_monoloco_rows(torch.zeros((0, 3, 17)), [synth.KITTI_K], np.zeros(0, dtype=np.int64)).shape == (0, 34)


def test_million_rows_addressing(hip_lib, cuda_device):
    """BASELINE config 4 in one piece (1,048,576 persons, 4.3 GB per activation buffer): every byte offset beyond
    2^32 must be computed in 64 bits.  Rows are independent, so slices taken at the far end of the batch must be
    bit-identical to the same rows run as a separate (tile-kernel) batch."""
    from monoloco_amd import engine
    m = 1 << 20
    eng = engine.LocoEngine({k: torch.tensor(v) for k, v in synth.make_state_dict(1).items()}, device=cuda_device)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    base = torch.tensor(synth.make_keypoints(65536, seed=21)).to(cuda_device)
    kps = base.repeat(16, 1, 1)
    kps += (torch.arange(m, device=cuda_device).view(-1, 1, 1) % 97).float() * 0.01   # every row distinct
    out, xyzds, raw = eng.forward_mono(kps, kinv, want_raw=True)
    assert torch.isfinite(raw).all()
    eng.set_tuning(mid_rows=0)      # the 2999-row reference batches on the tile kernels too
    for lo in (0, m // 2 + 12345, m - 3000):
        sub = slice(lo, lo + 2999)
        _, xyzds_s, raw_s = eng.forward_mono(kps[sub].contiguous(), kinv, want_raw=True)
        assert torch.equal(raw_s, raw[sub]) and torch.equal(xyzds_s, xyzds[sub]), lo
    eng.close()
    del out, xyzds, raw, kps
    torch.cuda.empty_cache()
