"""-m gpu: parity at the HEADLINE widths and sizes ("what's weak" 1 / "next round" 1).

* the reference's own fixture training at its default width 1024 (run.py:101; W-B-1024, checkpoints committed under
  tests/golden/ckpt_*_h1024.npz, outputs recorded from the real reference by oracle/make_golden.py wb1024);
* the hyper-parameter-search widths 512 / 2048 (train/hyp_tuning.py:52) and widths that are not a multiple of the
  256-column tile (600 mono, 200 stereo);
* BASELINE config 3 at its stated size (MonStereo 256 x 128 = 32768 pair rows) against an oracle sample;
* BASELINE config 5 at hidden 1024 (one training step against the fp32 and fp64 oracle);
* what happens to an activation beyond the fp16 range (the documented +-65504 clamp of the hi half).
"""
import os

import numpy as np
import pytest
import torch

import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


@pytest.fixture(scope='module')
def wb():
    return dict(np.load(os.path.join(G, 'golden_wb1024.npz')))


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(os.path.join(G, 'golden_path.npz')))


def _ckpt(mode):
    return {k: torch.tensor(v) for k, v in np.load(os.path.join(G, 'ckpt_%s_h1024.npz' % mode)).items()}


def _model(mode):
    from monoloco_amd.network.architectures import LocoModel
    sd = _ckpt(mode)
    m = LocoModel(68 if mode == 'stereo' else 34, 10 if mode == 'stereo' else 9, 1024)
    m.load_state_dict(sd, strict=False)
    return m


@pytest.mark.parametrize("merge", [True, False])
def test_wb1024_mono_reference_trained(hip_lib, cuda_device, wb, gold, merge):
    """500 fixture poses through the 1024-wide net the reference itself trained: raw outputs, the (x,y,z,d,sigma)
    parity tensor and every field of Loco.forward against the reference's results."""
    from monoloco_amd import engine
    eng = engine.LocoEngine(_ckpt('mono'), device=cuda_device, merge_w2w3=merge)
    conf = np.linspace(0.2, 1.0, 500).astype(np.float32)
    out, xyzds, raw = eng.forward_mono(torch.tensor(gold['mono_kps']), engine.inverse_intrinsics(synth.KITTI_K),
                                       box_conf=conf, want_raw=True)
    out, xyzds, raw = out.cpu().numpy(), xyzds.cpu().numpy(), raw.cpu().numpy()
    e_raw = np.abs(raw - wb['mono_raw']).max()
    e_raw64 = np.abs(raw - wb['mono_raw64']).max()
    noise = np.abs(wb['mono_raw'] - wb['mono_raw64']).max()
    ref_par = np.concatenate((wb['mono_xyz_pred'], wb['mono_d'], wb['mono_bi']), 1)
    e_par = np.abs(xyzds - ref_par).max()
    print("W-B-1024 mono (merge=%s): raw vs ref fp32 %.2e, vs ref fp64 %.2e (ref fp32-vs-fp64 %.2e), xyzds %.2e; d %.2f..%.2f m"
          % (merge, e_raw, e_raw64, noise, e_par, wb['mono_d'].min(), wb['mono_d'].max()))
    assert e_raw <= TOL and e_par <= TOL
    assert e_raw64 <= max(3 * noise, 2e-5)
    assert np.abs(out[:, 0:2] - wb['mono_xyzd'][:, 0:2]).max() <= TOL
    assert np.abs(out[:, 8:11] - np.concatenate((wb['mono_h'], wb['mono_w'], wb['mono_l']), 1)).max() <= TOL
    assert np.abs(out[:, 5] - wb['mono_yaw_pred'][:, 0]).max() <= TOL
    assert np.abs(out[:, 11] / wb['mono_conf'] - 1).max() <= 1e-4
    z_ref, d_ref = wb['mono_xyzd'][:, 2], wb['mono_xyzd'][:, 3]
    both = ~np.isnan(z_ref) & ~np.isnan(out[:, 2])
    amp = (np.abs(d_ref) / np.maximum(z_ref, 1e-3))[both]
    assert (np.abs(out[:, 2] - z_ref)[both] / np.maximum(amp, 1)).max() <= TOL
    eng.close()


def test_wb1024_mono_loco_surface(hip_lib, cuda_device, wb, gold):
    from monoloco_amd.network import Loco
    net = Loco(model=_model('mono'), mode='mono', device=cuda_device, linear_size=1024)
    dic = net.forward(gold['mono_kps'].tolist(), synth.KITTI_K)
    for key in ('h', 'w', 'l', 'ori', 'bi', 'd'):
        assert np.abs(dic[key].numpy() - wb['mono_' + key]).max() <= TOL, key
    assert np.abs(dic['yaw'][0].numpy() - wb['mono_yaw_pred']).max() <= TOL
    # yaw_ego = yaw + atan2(x, z) with the ill-conditioned spherical z = sqrt(d^2 - x^2 - y^2) (process.py:265):
    # a deviation dz <= TOL * d/z moves atan2 by |x| / (x^2 + z^2) * dz
    x, z, d = wb['mono_xyzd'][:, 0:1], wb['mono_xyzd'][:, 2:3], wb['mono_xyzd'][:, 3:4]
    ok = ~np.isnan(z)
    allowed = 2 * TOL + np.abs(x) / (x ** 2 + z ** 2) * TOL * d / np.maximum(z, 1e-3)
    err = np.abs(dic['yaw'][1].numpy() - wb['mono_yaw_ego'])
    err = np.minimum(err, np.abs(err - 2 * np.pi))          # the +-2 pi wrap may flip exactly at the branch cut
    assert (err[ok] <= allowed[ok]).all(), (err[ok] / allowed[ok]).max()


def test_wb1024_stereo_reference_trained(hip_lib, cuda_device, wb, gold):
    from monoloco_amd import engine
    from monoloco_amd.network import Loco
    eng = engine.LocoEngine(_ckpt('stereo'), device=cuda_device)
    raw = eng.forward_raw(torch.tensor(gold['stereo_x_fixture'])).cpu().numpy()
    e = np.abs(raw - wb['stereo_raw_fixture']).max()
    e64 = np.abs(raw - wb['stereo_raw64_fixture']).max()
    noise = np.abs(wb['stereo_raw_fixture'] - wb['stereo_raw64_fixture']).max()
    print("W-B-1024 stereo fixture rows: raw vs ref fp32 %.2e, vs fp64 %.2e (ref noise %.2e)" % (e, e64, noise))
    assert e <= TOL and e64 <= max(3 * noise, 2e-5)
    eng.close()
    nl, nr = (int(v) for v in wb['stereo_ava_nl_nr'])
    net = Loco(model=_model('stereo'), mode='stereo', device=cuda_device, linear_size=1024)
    dic = net.forward(gold['stereo_kps_l'][:nl].tolist(), synth.KITTI_K, keypoints_r=gold['stereo_kps_r'][:nr].tolist())
    ref_all = wb['stereo_ava_raw_all'].reshape(nl, nr, 10)
    gap = np.sort(ref_all[:, :, -1], 1)
    clear = (gap[:, -1] - gap[:, -2]) > 2 * TOL
    assert clear.sum() >= nl // 2
    for key in ('d', 'bi', 'h', 'w', 'l', 'aux'):
        assert np.abs(dic[key].numpy() - wb['stereo_ava_' + key])[clear].max() <= TOL, key


@pytest.mark.parametrize("seed,in_f,out_f,hidden", [(21, 34, 9, 512), (22, 34, 9, 2048), (23, 34, 9, 600), (24, 68, 10, 200)])
def test_other_widths(hip_lib, cuda_device, wb, gold, seed, in_f, out_f, hidden):
    """hyp_tuning.py:52 searches 512/1024/2048; the reference accepts any linear_size (600, 200: zero-padded here)."""
    from monoloco_amd import engine
    sd = synth.make_state_dict(seed, in_f, out_f, hidden)
    tag = 'w%d_' % hidden
    assert abs(synth.checksum(sd) - float(wb[tag + 'checksum'])) <= 1e-6 * abs(float(wb[tag + 'checksum']))
    x = torch.tensor(gold['mono_x_kitti']) if in_f == 34 else torch.tensor(gold['stereo_x_fixture'])
    for merge in (True, False):
        eng = engine.LocoEngine({k: torch.tensor(v) for k, v in sd.items()}, device=cuda_device, merge_w2w3=merge)
        raw = eng.forward_raw(x).cpu().numpy()
        e = np.abs(raw - wb[tag + 'raw']).max()
        e64 = np.abs(raw - wb[tag + 'raw64']).max()
        noise = np.abs(wb[tag + 'raw'] - wb[tag + 'raw64']).max()
        print("hidden %d (merge=%s): raw vs ref fp32 %.2e, vs fp64 %.2e (ref noise %.2e)" % (hidden, merge, e, e64, noise))
        assert e <= TOL and e64 <= max(3 * noise, 2e-5)
        # the tile kernel as well (a few hundred rows take the small-row kernels): replicate the rows
        big = eng.forward_raw(x.repeat(6, 1)[:3000]).cpu().numpy()
        assert np.abs(big[:len(x)] - wb[tag + 'raw']).max() <= TOL
        eng.close()


def test_config3_stereo_full_size(hip_lib, cuda_device, gold):
    """BASELINE config 3: MonStereo, 256 left x 128 right = 32768 pair rows in ONE call (K = 96 first layer, fused
    9-wide head, per-left arg-max); a 600-row sample against the CPU oracle, plus the selection itself."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    sd = _ckpt('stereo')
    eng = engine.LocoEngine(sd, device=cuda_device)
    rng = np.random.default_rng(3)
    kl = gold['stereo_kps_l'][rng.integers(0, 556, 256)] + rng.normal(0, 0.5, (256, 3, 17)).astype(np.float32)
    kr = gold['stereo_kps_r'][rng.integers(0, 556, 128)] + rng.normal(0, 0.5, (128, 3, 17)).astype(np.float32)
    kl, kr = torch.tensor(kl), torch.tensor(kr)
    res = eng.forward_stereo(kl, kr, engine.inverse_intrinsics(synth.KITTI_K), want_raw_all=True)
    raw_all = res['raw_all'].cpu()
    assert raw_all.shape == (32768, 10) and torch.isfinite(raw_all).all()
    idx = torch.arange(0, 32768, 53)[:600]
    x_l = O.preprocess_monoloco(kl, synth.KITTI_K)
    x_r = O.preprocess_monoloco(kr, synth.KITTI_K)
    li, ri = idx // 128, idx % 128
    rows = torch.cat((x_l[li], x_l[li] - x_r[ri]), 1)
    ref = O.loco_forward(sd, rows)
    ref64 = O.loco_forward(sd, rows.double(), dtype=torch.float64)
    e, noise = (raw_all[idx] - ref).abs().max().item(), (ref.double() - ref64).abs().max().item()
    print("config 3 (32768 pair rows): raw vs oracle fp32 %.2e, vs fp64 %.2e (oracle fp32-vs-fp64 %.2e)"
          % (e, (raw_all[idx].double() - ref64).abs().max().item(), noise))
    assert e <= TOL
    # per-left selection = arg-max of the aux logit over that left's 128 rows, computed here from the device's own rows
    best = res['best'].cpu().long()
    aux = raw_all[:, 9].reshape(256, 128)
    assert torch.equal(aux.gather(1, best.view(-1, 1)).reshape(-1), aux.max(1).values)
    chosen = raw_all.reshape(256, 128, 10)[torch.arange(256), best]
    assert (res['out'].cpu()[:, 3] - chosen[:, 2]).abs().max().item() == 0.0      # d of the packed result = chosen row's
    eng.close()


def test_config5_training_step_hidden_1024(hip_lib, cuda_device):
    """BASELINE config 5 at the reference's default width: one training step (dropout 0) on the 331-row fixture batch,
    losses / outputs / gradients against the fp32 and fp64 oracle."""
    from monoloco_amd.train import HipTrainer
    from oracle.train_oracle import OracleTrainer
    g = dict(np.load(os.path.join(G, 'golden_train_inputs.npz')))
    x, y = torch.tensor(g['mono_x']), torch.tensor(g['mono_y'])
    sd0 = {k: torch.tensor(v) for k, v in synth.make_state_dict(31, 34, 9, 1024).items()}
    tr = HipTrainer(sd0, p_dropout=0.0, lr=0.001, device=cuda_device)
    res, out = tr.step(x, y, update=False, want_outputs=True)
    o32 = OracleTrainer(sd0, lr=0.001)
    o64 = OracleTrainer(sd0, lr=0.001, dtype=torch.float64)
    l32, out32 = o32.step(x, y, update=False)
    l64, out64 = o64.step(x.double(), y.double(), update=False)
    e_out, e_out32 = (out.cpu().double() - out64).abs().max().item(), (out32.double() - out64).abs().max().item()
    print("config 5 train-mode outputs vs fp64: HIP %.2e, torch fp32 %.2e (|out| up to %.1f)" % (e_out, e_out32, out64.abs().max().item()))
    assert e_out <= max(4 * e_out32, 1e-4)     # batch-statistics BN at random init amplifies fp32 rounding; same class as torch
    assert abs(res['loss'] - l64['loss']) <= max(4 * abs(l32['loss'] - l64['loss']), 1e-5 * abs(l64['loss']))
    g_hip, g32, g64 = tr.grads(), o32.grads(), o64.grads()
    worst = 0.0
    for k in g_hip:
        scale = g64[k].abs().max().item() + 1e-12
        e_hip = (g_hip[k].double() - g64[k]).abs().max().item() / scale
        e_t32 = (g32[k].double() - g64[k]).abs().max().item() / scale
        worst = max(worst, e_hip)
        assert e_hip <= max(8 * e_t32, 2e-5), (k, e_hip, e_t32)
    print("config 5, hidden 1024: loss %.6f (oracle fp64 %.6f), worst relative gradient error %.2e" % (res['loss'], l64['loss'], worst))
    tr.close()


def test_activation_beyond_fp16_range_is_clamped_not_inf(hip_lib, cuda_device):
    """An fp32 reference has no ceiling at 65504; this path carries activations as fp16 hi+lo and CLAMPS the hi half
    to +-65504 (csrc/dense_kernel_pp.h split2_*).  Documented behaviour: inside the range results are fp32-class,
    beyond it the value saturates (finite, no inf/NaN), so an unnormalised input degrades instead of poisoning the row."""
    from monoloco_amd import engine
    rng = np.random.default_rng(5)
    k, n = 64, 256
    w = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    b = np.zeros(n, np.float32)
    x = rng.uniform(-1, 1, (300, k)).astype(np.float32)
    x[7] *= 3.0e4            # pre-activations of this row reach ~1e5 .. 3e5 > 65504
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    for small in (True, False):  # small-row kernels (16x16 and 32x32 tiles share the epilogue) and the tile kernel
        y = engine.debug_linear(torch.tensor(x).to(cuda_device), w, b, relu=False, small_path=small).cpu().numpy()
        assert np.isfinite(y).all()
        ok = np.abs(ref) < 6.0e4
        assert np.abs(y - ref)[ok].max() <= 2e-6 * np.abs(ref[ok]).max() + 1e-5
        sat = np.abs(ref) > 7.0e4
        assert sat[7].any() and not sat[np.arange(300) != 7].any()
        assert np.all(np.abs(y[sat]) <= 65504.0 * (1 + 2 ** -10)) and np.all(np.abs(y[sat]) >= 65504.0 * (1 - 2 ** -10))
        assert np.all(np.sign(y[sat]) == np.sign(ref[sat]))


def test_bf16_comparison_mode(hip_lib, cuda_device, wb, gold):
    """ML_PREC_BF16 (the precision BASELINE configs[1] literally names): runs, is deterministic, and -- measured, not
    assumed -- misses the 1e-4 bar by orders of magnitude on the reference-trained net, which is why it is a comparison
    mode and never the reported one.  A single bf16 dense layer is checked exactly against a bf16-operand emulation."""
    from monoloco_amd import engine
    # (a) one layer, exact model: bf16-rounded operands, fp32-class accumulation
    rng = np.random.default_rng(2)
    k, n, m = 96, 256, 700
    w = rng.uniform(-1, 1, (n, k)).astype(np.float32)
    b = rng.uniform(-1, 1, n).astype(np.float32)
    x = rng.uniform(-2, 2, (m, k)).astype(np.float32)
    y = engine.debug_linear(torch.tensor(x).to(cuda_device), w, b, relu=True, precision='bf16').cpu().numpy()
    rb = lambda a: torch.tensor(a).to(torch.bfloat16).to(torch.float64).numpy()
    # the packer scales W by a power of two before rounding (exact), so rounding commutes with it
    ref = np.maximum(rb(x) @ rb(w).T + b.astype(np.float64), 0.0)
    ref = rb(ref.astype(np.float32))                      # the layer's output is stored as bf16
    assert np.abs(y - ref).max() <= 2 ** -7 * np.abs(ref).max()   # at most one bf16 ulp (fp32 vs fp64 accumulation at a tie)
    assert (y != ref).mean() < 0.01
    # (b) the whole net on the reference-trained 1024-wide weights
    eng = engine.LocoEngine(_ckpt('mono'), device=cuda_device, precision='bf16')
    kps = torch.tensor(np.tile(gold['mono_kps'], (6, 1, 1))[:2560])
    _, xyzds, raw = eng.forward_mono(kps, engine.inverse_intrinsics(synth.KITTI_K), want_raw=True)
    _, xyzds2, _ = eng.forward_mono(kps, engine.inverse_intrinsics(synth.KITTI_K))
    assert torch.equal(xyzds, xyzds2)
    raw = raw.cpu().numpy()[:500]
    e = np.abs(raw - wb['mono_raw64']).max(0)
    print("bf16 mode on W-B-1024: max |raw - ref fp64| per column", np.array2string(e, precision=3))
    assert np.isfinite(raw).all()
    assert e[2] > 1e-3, "bf16 meeting the bar would make the 3-product mode pointless: re-measure"
    assert e[2] < 5.0          # still the same function: distance errors of centimetres to decimetres, not garbage
    eng.close()


@pytest.mark.parametrize("mode", ["mono", "stereo"])
def test_reference_trained_weights_at_the_full_batch(hip_lib, cuda_device, gold, mode):
    """BASELINE configs[1] / [2] at their full sizes (65536 persons; 256 x 128 = 32768 pair rows) on the REFERENCE-TRAINED 1024-wide
    checkpoints (round-3 review, weak 1c: they had only met 500 / 556 rows): SURVEY 8d's parity set P -- the fixture's real poses tiled
    to the batch with a per-row jitter (u, v += N(0, 0.5 px)) -- against the CPU oracle on EVERY row (round 5) at the north-star
    tolerance, plus the size-independent property that every row equals the same row computed in a small batch of its own (another
    kernel family: small-row / mid-size path)."""
    from monoloco_amd import engine
    from oracle import monoloco_oracle as O
    sd = _ckpt(mode)
    kinv = engine.inverse_intrinsics(synth.KITTI_K)
    rng = np.random.default_rng(11)
    eng = engine.LocoEngine(sd, device=cuda_device)
    if mode == 'mono':
        m = 65536
        base = gold['mono_kps']
        kps = base[rng.integers(0, len(base), m)].copy()
        kps[:, 0:2, :] += rng.normal(0, 0.5, (m, 2, 17)).astype(np.float32)
        kt = torch.tensor(kps).to(cuda_device)
        conf = torch.tensor(rng.random(m).astype(np.float32)).to(cuda_device)
        out, xyzds, raw = eng.forward_mono(kt, kinv, box_conf=conf, want_raw=True)
        # round 5: EVERY row of the batch against the oracle (8192-row chunks, a few seconds), not a strided sample
        raw_h, xyzds_h, conf_h = raw.cpu(), xyzds.cpu(), conf.cpu()
        d_lo, d_hi = 1e9, 0.0
        for lo in range(0, m, 8192):
            ref = O.forward_mono(sd, torch.tensor(kps[lo:lo + 8192]), synth.KITTI_K, box_conf=conf_h[lo:lo + 8192])
            assert (raw_h[lo:lo + 8192] - ref['raw']).abs().max().item() <= TOL, lo
            assert (xyzds_h[lo:lo + 8192] - ref['xyzds']).abs().max().item() <= TOL, lo
            d_lo, d_hi = min(d_lo, ref['raw'][:, 2].min().item()), max(d_hi, ref['raw'][:, 2].max().item())
        assert d_lo > 0.3 and d_hi > 15.0                             # a trained net on real poses: metres, not noise
        idx = np.arange(0, m, m // 512)[:512]
        sub = idx[:300]
        out_s, xyzds_s, raw_s = eng.forward_mono(kt[sub], kinv, box_conf=conf[sub], want_raw=True)    # 300 rows: the small-row kernels
        assert (raw_s - raw[sub]).abs().max().item() <= 4e-6 * max(1.0, raw.abs().max().item())
    else:
        ml, mr = 256, 128
        kl = gold['stereo_kps_l'][rng.integers(0, len(gold['stereo_kps_l']), ml)].copy()
        kr = gold['stereo_kps_r'][rng.integers(0, len(gold['stereo_kps_r']), mr)].copy()
        kl[:, 0:2, :] += rng.normal(0, 0.5, (ml, 2, 17)).astype(np.float32)
        kr[:, 0:2, :] += rng.normal(0, 0.5, (mr, 2, 17)).astype(np.float32)
        res = eng.forward_stereo(torch.tensor(kl).to(cuda_device), torch.tensor(kr).to(cuda_device), kinv, want_raw_all=True)
        raw_all = res['raw_all'].cpu()
        assert raw_all.shape == (ml * mr, 10)
        x, _ = O.preprocess_monstereo(torch.tensor(kl), torch.tensor(kr), synth.KITTI_K)
        for lo in range(0, ml * mr, 8192):      # every pair row (round 5; a strided 512-row sample before)
            ref = O.loco_forward(sd, x[lo:lo + 8192])
            err = (raw_all[lo:lo + 8192] - ref).abs()
            assert err[:, :9].max().item() <= TOL, lo
            # the aux logit of a non-matching pair is -50 .. -90 on this checkpoint (its sigmoid is 0 to 20+ digits): fp32 holds such
            # a value to 4-8e-6, the reference's own fp32 run is that far from fp64 -- the absolute bar plus 4e-6 relative
            assert (err[:, 9] <= TOL + 4e-6 * ref[:, 9].abs()).all(), lo
            assert (torch.sigmoid(raw_all[lo:lo + 8192, 9]) - torch.sigmoid(ref[:, 9])).abs().max().item() <= TOL
        # the per-left winner is the arg-max of the device's own aux logits
        best = res['best'].cpu().long()
        assert torch.equal(best, raw_all.view(ml, mr, 10)[:, :, -1].argmax(1))
    eng.close()
