"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference; the GPU box never sees it).  The
reference is imported read-only with stub modules for its unused heavy imports (torchvision is
imported at monoloco/network/process.py:9 but only used by image_transform).  Nothing from the
reference is copied: this script calls it and stores inputs/outputs as .npz / .json fixtures.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
"""
import argparse
import copy
import json
import os
import sys
import types

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
for name in ('torchvision', 'torchvision.transforms'):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from monoloco.network import Loco, load_calibration, preprocess_pifpaf  # noqa: E402
from monoloco.network.architectures import LocoModel  # noqa: E402
from monoloco.network.process import preprocess_monoloco, preprocess_monstereo  # noqa: E402
from monoloco.train import Trainer  # noqa: E402
from monoloco.utils import get_keypoints, pixel_to_camera, xyz_from_distance  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def np_sd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def ref_model(sd, in_f, out_f, hidden, dtype=torch.float32):
    m = LocoModel(in_f, out_f, hidden, device='cpu')
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    m.eval()
    return m.to(dtype)


def train(mode, hidden, epochs, tmp):
    """The reference's own fixture training (tests/test_train_{mono,stereo}.py: lr 0.001, -e 10 / 20)."""
    args = argparse.Namespace(mode=mode, joints=os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode),
                              epochs=epochs, no_save=True, print_loss=False, lr=0.001, sched_step=30,
                              sched_gamma=0.98, hidden_size=hidden, n_stage=3, r_seed=1, auto_tune_mtl=False,
                              out=os.path.join(tmp, 'x.pkl'), bs=512, dropout=0.2)
    tr = Trainer(args)
    tr.train()
    return {k: v.clone() for k, v in tr.model.state_dict().items()}


def dic_to_np(dic, prefix):
    out = {}
    for k, v in dic.items():
        if k == 'yaw':
            out[prefix + 'yaw_pred'] = v[0].numpy()
            out[prefix + 'yaw_ego'] = v[1].numpy()
        elif k == 'epi':
            out[prefix + 'epi'] = np.asarray(v, dtype=np.float32)
        else:
            out[prefix + k] = v.numpy()
    return out


def geometry(kps, kk, d, bi, conf):
    """The per-person geometry of Loco.post_process, through the reference's own functions."""
    uv_c = get_keypoints(kps, mode='center')
    xy_c = pixel_to_camera(uv_c, kk, 1)
    xyz = xyz_from_distance(d, xy_c)
    dist = torch.sqrt((xyz.double() ** 2).sum(1))
    cf = 0.035 * torch.as_tensor(conf).double() / (bi.reshape(-1).double() / dist)
    return xyz.numpy(), cf.numpy()


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = '/tmp/make_golden'
    os.makedirs(tmp, exist_ok=True)
    torch.set_num_threads(8)
    g = {}

    # ---------------------------------------------------------------- weights
    sd_a = synth.make_state_dict(1, 34, 9, 1024)           # W-A mono  (seeded synthetic, 1024)
    sd_as = synth.make_state_dict(3, 68, 10, 1024)         # W-A stereo
    g['synth_checksum_mono'] = np.float64(synth.checksum(sd_a))
    g['synth_checksum_stereo'] = np.float64(synth.checksum(sd_as))
    sd_b = np_sd(train('mono', 256, 10, tmp))              # W-B mono  (reference fixture training, 256)
    sd_bs = np_sd(train('stereo', 256, 20, tmp))           # W-B stereo
    np.savez(os.path.join(OUT, 'ckpt_mono_h256.npz'), **sd_b)
    np.savez(os.path.join(OUT, 'ckpt_stereo_h256.npz'), **sd_bs)

    # ---------------------------------------------------------------- fixture poses (sample_joints-*.json)
    dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-mono.json')))
    kps = torch.tensor(dj['train']['kps'] + dj['val']['kps'])[:, 0]        # (500, 3, 17)
    x_fix = torch.tensor(dj['train']['X'] + dj['val']['X'])               # (500, 34) stored by the reference prep
    ks = []
    for k in dj['train']['K'] + dj['val']['K']:
        kt = torch.tensor(k)
        if not any(torch.equal(kt, u) for u in ks):
            ks.append(kt)
    g['mono_kps'] = kps.numpy()
    g['mono_x_fixture'] = x_fix.numpy()
    g['mono_unique_k'] = torch.stack(ks).numpy()
    # which K reproduces each fixture row bit-exactly (the data pin of the pre-process)
    which = np.full(len(kps), -1)
    for i, k in enumerate(ks):
        hit = (preprocess_monoloco(kps, k) == x_fix).all(1).numpy()
        which[(which < 0) & hit] = i
    assert (which >= 0).all()
    g['mono_k_index'] = which
    kk = synth.KITTI_K
    g['kk'] = np.asarray(kk, dtype=np.float64)
    conf = np.linspace(0.2, 1.0, len(kps)).astype(np.float32)
    g['mono_conf'] = conf
    for tag, sd, hidden in (('A', sd_a, 1024), ('B', sd_b, 256)):
        with torch.no_grad():
            x = preprocess_monoloco(kps, torch.tensor(kk))
            net = Loco(model=ref_model(sd, 34, 9, hidden), mode='mono', linear_size=hidden)
            dic = net.forward(kps.tolist(), kk)
            raw = net.model(x)
            raw64 = ref_model(sd, 34, 9, hidden, torch.float64)(
                preprocess_monoloco(kps.double(), torch.tensor(kk, dtype=torch.float64)))
        g.update(dic_to_np(dic, 'mono_%s_' % tag))
        g['mono_%s_raw' % tag] = raw.numpy()
        g['mono_%s_raw64' % tag] = raw64.numpy()
        xyz, cf = geometry(kps, kk, dic['d'], dic['bi'], conf)
        g['mono_%s_xyz_pred' % tag] = xyz
        g['mono_%s_conf' % tag] = cf
        if tag == 'A':
            g['mono_x_kitti'] = x.numpy()

    # ---------------------------------------------------------------- stereo fixture pairs
    ds = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-stereo.json')))
    kps_s = torch.tensor(ds['train']['kps'] + ds['val']['kps'])[:, 0]     # (556, 3, 34): left 17 | right 17
    x_s = torch.tensor(ds['train']['X'] + ds['val']['X'])                 # (556, 68)
    kl, kr = kps_s[:, :, :17].contiguous(), kps_s[:, :, 17:].contiguous()
    ks_s = []
    for k in ds['train']['K'] + ds['val']['K']:
        kt = torch.tensor(k)
        if not any(torch.equal(kt, u) for u in ks_s):
            ks_s.append(kt)
    which = np.full(len(kl), -1)
    for i, k in enumerate(ks_s):
        xl, xr = preprocess_monoloco(kl, k), preprocess_monoloco(kr, k)
        hit = (torch.cat((xl, xl - xr), 1) == x_s).all(1).numpy()
        which[(which < 0) & hit] = i
    g['stereo_kps_l'] = kl.numpy()
    g['stereo_kps_r'] = kr.numpy()
    g['stereo_x_fixture'] = x_s.numpy()
    g['stereo_unique_k'] = torch.stack(ks_s).numpy()
    g['stereo_k_index'] = which
    for tag, sd, hidden in (('A', sd_as, 1024), ('B', sd_bs, 256)):
        with torch.no_grad():
            raw = ref_model(sd, 68, 10, hidden)(x_s)
            raw64 = ref_model(sd, 68, 10, hidden, torch.float64)(x_s.double())
        g['stereo_%s_raw_fixture' % tag] = raw.numpy()
        g['stereo_%s_raw64_fixture' % tag] = raw64.numpy()
        # all-vs-all through Loco.forward: 40 left x 7 right fixture poses
        nl, nr = 40, 7
        with torch.no_grad():
            net = Loco(model=ref_model(sd, 68, 10, hidden), mode='stereo', linear_size=hidden)
            dic = net.forward(kl[:nl].tolist(), kk, keypoints_r=kr[:nr].tolist())
            inputs, _ = preprocess_monstereo(kl[:nl], kr[:nr], torch.tensor(kk))
            raw_all = net.model(inputs)
        g.update(dic_to_np(dic, 'stereo_%s_ava_' % tag))
        g['stereo_%s_ava_raw_all' % tag] = raw_all.numpy()
        if tag == 'A':
            g['stereo_ava_inputs'] = inputs.numpy()
            # no right keypoints: right := first left pose (net.py:115-116)
            with torch.no_grad():
                dic0 = net.forward(kl[:5].tolist(), kk)
            g.update(dic_to_np(dic0, 'stereo_A_noright_'))
    g['stereo_ava_nl_nr'] = np.array([40, 7])

    np.savez_compressed(os.path.join(OUT, 'golden_path.npz'), **g)

    # ---------------------------------------------------------------- C1: the pifpaf fixture image
    ann = json.load(open(os.path.join(REF, 'tests', '002282.png.pifpaf.json')))
    json.dump(ann, open(os.path.join(OUT, 'pifpaf_002282.json'), 'w'))
    c1 = {}
    for name, kwargs in (('predict', dict(im_size=(1238, 374), enlarge_boxes=False)),
                         ('default', dict(im_size=None)), ('eval', dict(im_size=(1242, 374))),
                         ('minconf', dict(im_size=(1238, 374), min_conf=0.55))):
        boxes, kpl = preprocess_pifpaf(copy.deepcopy(ann), **kwargs)
        c1['pre_' + name] = {'boxes': boxes, 'keypoints': kpl}
    # a score-keyed annotation (bbox as x, y, w, h)
    ann_s = copy.deepcopy(ann[:3])
    for i, a in enumerate(ann_s):
        b = a['bbox']
        a['bbox'] = [b[0], b[1], b[2] - b[0], b[3] - b[1]]
        a['score'] = 0.5 + 0.1 * i
    boxes, kpl = preprocess_pifpaf(copy.deepcopy(ann_s), im_size=(1238, 374))
    c1['pre_score'] = {'input': ann_s, 'boxes': boxes, 'keypoints': kpl}
    c1['calib'] = {'kitti_1238_374': load_calibration('kitti', (1238, 374)),
                   'kitti_1242_375': load_calibration('kitti', (1242, 375)),
                   'custom_1920_1080': load_calibration('custom', (1920, 1080), focal_length=5.7),
                   'nuscenes_1600_900': load_calibration('nuscenes', (1600, 900))}
    boxes, kpl = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
    kk1 = load_calibration('kitti', (1238, 374))
    c1np = {}
    for tag, sd, hidden in (('A', sd_a, 1024), ('B', sd_b, 256)):
        net = Loco(model=ref_model(sd, 34, 9, hidden), mode='mono', linear_size=hidden)
        dic = net.forward(kpl, kk1)
        c1np.update(dic_to_np(dic, 'fwd_%s_' % tag))
        pp = Loco.post_process(dic, boxes, kpl, kk1)
        c1['post_%s' % tag] = dict(pp)
        # with ground truth: fake gt = a few of the detections' own boxes, shuffled distances
        dic_gt = {'boxes': [boxes[i][:4] for i in (3, 0, 7, 12)],
                  'ys': [[0, 0, 0, 10.0 + 2 * i] for i in range(4)]}
        c1['post_gt_%s' % tag] = dict(Loco.post_process(dic, boxes, kpl, kk1, dic_gt=dic_gt))
        c1['dic_gt'] = dic_gt
    np.savez_compressed(os.path.join(OUT, 'golden_c1.npz'), **c1np)
    json.dump(c1, open(os.path.join(OUT, 'golden_c1.json'), 'w'))

    # the reference's own unit test of this path (tests/test_utils.py:18-25): exact linearity in z_met
    uv = [1000., 400.]
    a = pixel_to_camera(uv, kk, 1)[0] * 10
    b = pixel_to_camera(uv, kk, 10)[0]
    assert a == b
    print('golden files written to', OUT)
    for f in sorted(os.listdir(OUT)):
        print('  %-28s %8.1f KiB' % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def golden_train():
    """Three iterations of the reference's own training loop body (trainer.py:150-161) with dropout 0 on its
    mono and stereo fixtures, from seeded weights: per-step losses, the clipped gradients of step 1, the state
    after step 3."""
    import itertools
    from monoloco.train.losses import CompositeLoss, MultiTaskLoss
    g = {}
    for mode, in_f, out_f, seed in (('mono', 34, 9, 7), ('stereo', 68, 10, 8)):
        hidden = 128
        dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode)))
        x = torch.tensor(dj['train']['X'])
        y = torch.tensor(dj['train']['Y'])
        tasks = ('d', 'x', 'y', 'h', 'w', 'l', 'ori') + (('aux',) if mode == 'stereo' else ())
        losses_tr, losses_val = CompositeLoss(tasks)()
        mt = MultiTaskLoss(losses_tr, losses_val, (1,) * len(tasks), tasks)
        model = LocoModel(in_f, out_f, hidden, p_dropout=0.0, device='cpu')
        model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.make_state_dict(seed, in_f, out_f, hidden).items()},
                              strict=False)
        model.train()
        opt = torch.optim.Adam(params=itertools.chain(model.parameters(), mt.parameters()), lr=0.001)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)   # small step so that the decay is exercised
        for step in range(3):
            opt.zero_grad()
            out = model(x)
            loss, vals = mt(out, y, phase='train')
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
            if step == 0:
                g[mode + '_out0'] = out.detach().numpy()
                for k, p in model.named_parameters():
                    g[mode + '_grad0/' + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy().copy()
            opt.step()
            sched.step()
            g[mode + '_loss%d' % step] = np.array([float(loss)] + [float(v) for v in vals])
        for k, v in model.state_dict().items():
            if v.dtype.is_floating_point:
                g[mode + '_final/' + k] = v.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'golden_train.npz'), **g)
    # the batches themselves (train/val split X, Y of the reference's fixtures), inputs of these goldens
    inp = {}
    for mode in ('mono', 'stereo'):
        dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode)))
        for ph, tag in (('train', ''), ('val', 'val')):
            inp[mode + '_x' + tag] = np.asarray(dj[ph]['X'], dtype=np.float32)
            inp[mode + '_y' + tag] = np.asarray(dj[ph]['Y'], dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, 'golden_train_inputs.npz'), **inp)
    print('golden_train.npz %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'golden_train.npz')) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train':
    golden_train()


def golden_train_big():
    """Two iterations of the reference's training loop body on LARGE batches (4096 / 5000 rows drawn from its fixtures,
    synth.big_train_batch), hidden 256, dropout 0: the regime in which the HIP step runs its hidden-layer GEMMs on the
    3-product fp16 MFMA kernel.  First-step outputs, losses of both steps, first-step (clipped) gradients: all of them for
    mono, the large matrices of one stage + the narrow layers for stereo."""
    import itertools
    from monoloco.train.losses import CompositeLoss, MultiTaskLoss
    g = {}
    for mode, in_f, out_f, seed, m in (('mono', 34, 9, 41, 4096), ('stereo', 68, 10, 42, 5000)):
        hidden = 256
        dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode)))
        xb, yb = synth.big_train_batch(np.asarray(dj['train']['X'], dtype=np.float32), np.asarray(dj['train']['Y'], dtype=np.float32),
                                       m, seed)
        x, y = torch.tensor(xb), torch.tensor(yb)
        tasks = ('d', 'x', 'y', 'h', 'w', 'l', 'ori') + (('aux',) if mode == 'stereo' else ())
        losses_tr, losses_val = CompositeLoss(tasks)()
        mt = MultiTaskLoss(losses_tr, losses_val, (1,) * len(tasks), tasks)
        model = LocoModel(in_f, out_f, hidden, p_dropout=0.0, device='cpu')
        model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.make_state_dict(seed, in_f, out_f, hidden).items()},
                              strict=False)
        model.train()
        opt = torch.optim.Adam(params=itertools.chain(model.parameters(), mt.parameters()), lr=0.001)
        keep = None if mode == 'mono' else ('w1.weight', 'linear_stages.1.w1.weight', 'linear_stages.1.w2.weight',
                                            'linear_stages.1.batch_norm2.weight', 'w2.weight', 'w3.weight', 'w_fin.weight',
                                            'w_aux.weight', 'w2.bias')
        for step in range(2):
            opt.zero_grad()
            out = model(x)
            loss, vals = mt(out, y, phase='train')
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
            if step == 0:
                g[mode + '_out0'] = out.detach().numpy()
                for k, p in model.named_parameters():
                    if keep is None or k in keep:
                        g[mode + '_grad0/' + k] = p.grad.numpy().copy()
            opt.step()
            g[mode + '_loss%d' % step] = np.array([float(loss)] + [float(v) for v in vals])
        g[mode + '_rows_seed'] = np.array([m, seed])
    np.savez_compressed(os.path.join(OUT, 'golden_train_big.npz'), **g)
    print('golden_train_big.npz %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'golden_train_big.npz')) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train_big':
    golden_train_big()


def golden_train_h1024():
    """The reference's training loop body (trainer.py:150-161) at the HEADLINE width (hidden 1024, run.py:101) on a 512-row
    batch (the reference's default --bs, run.py:95: the HIP step's "mid" route) and a 4096-row batch (its large-batch
    route), dropout 0, from seeded weights: first-step outputs, losses of two steps, first-step clipped gradients -- every
    narrow tensor in full, of each 1024 x 1024 matrix every 64th row (16 x 1024 values), the row and column sums of all 1024
    rows / columns, one matrix in full, and the tensor's max |g|.  The same
    run in fp64 gives the reference's own fp32 rounding noise per tensor ('_noise/<key>' = max |g32 - g64| / max |g64|), the
    yardstick the GPU test's tolerances are set by."""
    import itertools
    from monoloco.train.losses import CompositeLoss, MultiTaskLoss
    g = {}
    mode, in_f, out_f, hidden = 'mono', 34, 9, 1024
    dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode)))
    for m, seed in ((512, 51), (4096, 52)):
        tag = 'r%d' % m
        xb, yb = synth.big_train_batch(np.asarray(dj['train']['X'], dtype=np.float32), np.asarray(dj['train']['Y'], dtype=np.float32),
                                       m, seed)
        tasks = ('d', 'x', 'y', 'h', 'w', 'l', 'ori')
        grads = {}
        for dtype in (torch.float32, torch.float64):
            x, y = torch.tensor(xb).to(dtype), torch.tensor(yb).to(dtype)
            losses_tr, losses_val = CompositeLoss(tasks)()
            mt = MultiTaskLoss(losses_tr, losses_val, (1,) * len(tasks), tasks)
            model = LocoModel(in_f, out_f, hidden, p_dropout=0.0, device='cpu')
            model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.make_state_dict(seed, in_f, out_f, hidden).items()},
                                  strict=False)
            model.to(dtype).train()
            opt = torch.optim.Adam(params=itertools.chain(model.parameters(), mt.parameters()), lr=0.001)
            for step in range(2 if dtype == torch.float32 else 1):
                opt.zero_grad()
                out = model(x)
                loss, vals = mt(out, y, phase='train')
                loss.backward()
                torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
                if step == 0:
                    grads[dtype] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
                    if dtype == torch.float32:
                        g[tag + '_out0'] = out.detach().numpy()
                    else:
                        g[tag + '_out0_f64'] = out.detach().numpy()
                opt.step()
                if dtype == torch.float32:
                    g[tag + '_loss%d' % step] = np.array([float(loss)] + [float(v) for v in vals])
        for k, v in grads[torch.float32].items():
            v64 = grads[torch.float64][k]
            g[tag + '_gmax/' + k] = np.array(v64.abs().max().item())
            g[tag + '_noise/' + k] = np.array(((v.double() - v64).abs().max() / v64.abs().max().clamp_min(1e-300)).item())
            if v.dim() == 2 and v.shape[0] == hidden and v.shape[1] == hidden:
                g[tag + '_grad0/' + k] = v[::64].numpy().copy()
                g[tag + '_grad0_f64/' + k] = v64[::64].float().numpy().copy()
                # full-matrix coverage that stays small (round 4): row and column sums of ALL 1024 rows / columns (fp64 sums of
                # the fp32 gradient, and of the fp64 run's) -- a wrong 32 x 64 tile anywhere moves 32 row sums and 64 column sums
                g[tag + '_rowsum/' + k] = v.double().sum(1).numpy().copy()
                g[tag + '_colsum/' + k] = v.double().sum(0).numpy().copy()
                g[tag + '_rowsum64/' + k] = v64.sum(1).numpy().copy()
                g[tag + '_colsum64/' + k] = v64.sum(0).numpy().copy()
                g[tag + '_rowabs64/' + k] = v64.abs().sum(1).numpy().copy()     # the scale a sum's error is judged against
                g[tag + '_colabs64/' + k] = v64.abs().sum(0).numpy().copy()
                if tag == 'r512' and k == 'linear_stages.1.w1.weight':           # ... and ONE matrix in full
                    g[tag + '_full/' + k] = v.numpy().copy()
            else:
                g[tag + '_grad0/' + k] = v.numpy().copy()
                g[tag + '_grad0_f64/' + k] = v64.float().numpy().copy()
            # the reference's own fp32 run against its fp64 run, as an rms ratio (the yardstick beside '_noise', which is the max)
            g[tag + '_noise_rms/' + k] = np.array(((v.double() - v64).pow(2).mean().sqrt() / v64.pow(2).mean().sqrt().clamp_min(1e-300)).item())
        g[tag + '_rows_seed'] = np.array([m, seed])
        print(tag, 'loss', g[tag + '_loss0'][0], 'noise: max', max(float(v) for k, v in g.items() if k.startswith(tag + '_noise/')))
    np.savez_compressed(os.path.join(OUT, 'golden_train_h1024.npz'), **g)
    print('golden_train_h1024.npz %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'golden_train_h1024.npz')) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train_h1024':
    golden_train_h1024()


def golden_train_autotune():
    """Three iterations of the reference's loop body with its AutoTuneMultiTaskLoss (`--auto_tune_mtl`, losses.py:17-43,
    trainer.py:95-96) on the mono and stereo fixtures, hidden 128, dropout 0: per step the total and the (weighted) task
    values, the log_sigmas after every step, the first step's outputs and log_sigma gradient."""
    import itertools
    from monoloco.train.losses import AutoTuneMultiTaskLoss, CompositeLoss
    g = {}
    for mode, in_f, out_f, seed in (('mono', 34, 9, 7), ('stereo', 68, 10, 8)):
        hidden = 128
        dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-%s.json' % mode)))
        x = torch.tensor(dj['train']['X'])
        y = torch.tensor(dj['train']['Y'])
        tasks = ('d', 'x', 'y', 'h', 'w', 'l', 'ori') + (('aux',) if mode == 'stereo' else ())
        losses_tr, losses_val = CompositeLoss(tasks)()
        mt = AutoTuneMultiTaskLoss(losses_tr, losses_val, (1,) * len(tasks), tasks)
        model = LocoModel(in_f, out_f, hidden, p_dropout=0.0, device='cpu')
        model.load_state_dict({k: torch.as_tensor(v) for k, v in synth.make_state_dict(seed, in_f, out_f, hidden).items()},
                              strict=False)
        model.train()
        opt = torch.optim.Adam(params=itertools.chain(model.parameters(), mt.parameters()), lr=0.001)
        sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
        for step in range(3):
            opt.zero_grad()
            out = model(x)
            loss, vals = mt(out, y, phase='train')
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 3)
            if step == 0:
                g[mode + '_out0'] = out.detach().numpy()
                g[mode + '_grad0_log_sigmas'] = mt.log_sigmas.grad.numpy().copy()
                g[mode + '_grad0_w1'] = model.w1.weight.grad.numpy().copy()
            opt.step()
            sched.step()
            g[mode + '_loss%d' % step] = np.array([float(loss)] + [float(v) for v in vals])
            g[mode + '_log_sigmas%d' % step] = mt.log_sigmas.detach().numpy().copy()
        _, vals_val = mt(model(x), y, phase='val')
        g[mode + '_val_tail'] = np.array([float(v) for v in vals_val[len(tasks):]])     # the sigmas appended in 'val' phase
    np.savez_compressed(os.path.join(OUT, 'golden_train_autotune.npz'), **g)
    print('golden_train_autotune.npz %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'golden_train_autotune.npz')) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'train_autotune':
    golden_train_autotune()


def golden_formats():
    """On-disk formats: the reference's preprocess_pifpaf on JSON texts (fixture + a synthetic x,y,w,h/'score'
    list) and the reference's save_txts (eval/generate_kitti.py:202-253) for every `net` branch on seeded
    synthetic outputs.  Stored as JSON (Python's float repr round-trips doubles exactly)."""
    os.makedirs('/tmp/make_golden/data/logs', exist_ok=True)
    os.chdir('/tmp/make_golden')  # monoloco.eval asserts that data/logs exists relative to the cwd
    sys.modules.setdefault('torchvision.models', types.ModuleType('torchvision.models'))
    from monoloco.eval.generate_kitti import save_txts
    gold = {'pifpaf': [], 'kitti': []}
    fixture = open(os.path.join(REF, 'tests', '002282.png.pifpaf.json')).read()
    rng = np.random.default_rng(5)
    synth_anns = []
    for i in range(23):
        kp = []
        for _ in range(17):
            kp += [float(rng.uniform(-20, 1300)), float(rng.uniform(-20, 400)), float(rng.uniform(0, 1))]
        ann = {'keypoints': kp, 'bbox': [float(rng.uniform(-30, 1200)), float(rng.uniform(-30, 350)),
                                          float(rng.uniform(1, 200)), float(rng.uniform(1, 300))],
               'score': float(rng.uniform(0, 1)), 'category_id': 1, 'extra': {'a': [1, {"b": 'x"y\\z ]}'}], 'c': None}}
        if i % 5 == 0:
            ann['bbox'] = [int(v) for v in ann['bbox']]  # ints in the JSON
        if i % 7 == 0:
            ann['score'] = 1
        synth_anns.append(ann)
    synth_text = json.dumps(synth_anns, indent=1)
    texts = {'fixture': fixture, 'synthetic_score': synth_text, 'empty': '[]'}
    gold['texts'] = {'synthetic_score': synth_text, 'empty': '[]'}
    for name, text in texts.items():
        for kw in (dict(), dict(im_size=[1238, 374], enlarge_boxes=False), dict(im_size=[1238, 374], min_conf=0.45),
                   dict(enlarge_boxes=False, min_conf=0.2)):
            boxes, kps = preprocess_pifpaf(json.loads(text), **{k: (tuple(v) if isinstance(v, list) else v)
                                                              for k, v in kw.items()})
            gold['pifpaf'].append({'text': name, 'kwargs': kw, 'boxes': boxes, 'keypoints': kps})

    torch.manual_seed(11)
    m = 9
    boxes = [[float(rng.uniform(0, 600)), float(rng.uniform(0, 200)), float(rng.uniform(600, 1238)),
              float(rng.uniform(200, 374)), float(rng.uniform(0.05, 1))] for _ in range(m)]
    xyzd = torch.randn(m, 4) * torch.tensor([5., 1., 10., 1.]) + torch.tensor([0., 1., 20., 20.])
    bis = torch.rand(m, 1) * 2 + 0.1
    epis = [0.] * (m - 3) + [float(v) for v in torch.rand(3)]
    yaws = (torch.rand(m, 1) * 6 - 3, torch.rand(m, 1) * 6 - 3)
    hs, ws, ls = torch.rand(m, 1) + 1, torch.rand(m, 1), torch.rand(m, 1)
    dds = torch.rand(m, 1) * 30 + 1
    xy_centers = torch.cat((torch.randn(m, 2) * 0.3, torch.ones(m, 1)), dim=1)
    zzs_geom = [float(v) for v in torch.rand(m) * 30 + 1]
    cat = [0.0, 1.0, 0.05, 0.2, 0.0, 0.0, 1.0, 0.0, 0.099][:m]
    tt = [0.06, -0.01, 0.003]
    xyz_b = [[float(v) for v in row] for row in torch.randn(m, 3) * 4 + torch.tensor([0., 1., 15.])]
    cases = {
        'monoloco_pp': ([xyzd, bis, epis, yaws, hs, ws, ls], [None, None]),
        'monstereo': ([xyzd, bis, epis, yaws, hs, ws, ls], [None, None]),
        'monoloco': ([dds, bis, epis, zzs_geom, xy_centers], [None, None]),
        'geometric': ([dds, bis, epis, zzs_geom, xy_centers], [None, None]),
        'baseline': ([xyz_b, bis, epis, zzs_geom, xy_centers], [None, tt]),
    }

    def plain(x):
        if isinstance(x, torch.Tensor):
            return {'tensor': x.tolist()}
        if isinstance(x, tuple):
            return {'tuple': [plain(v) for v in x]}
        return x

    for net, (outs, params) in cases.items():
        path = os.path.join('/tmp/make_golden', 'kitti_%s.txt' % net)
        save_txts(path, copy.deepcopy(boxes), outs, params, net=net, cat=cat)
        gold['kitti'].append({'net': net, 'boxes': boxes, 'outputs': [plain(o) for o in outs], 'params': params,
                              'cat': cat, 'text': open(path).read()})
    with open(os.path.join(OUT, 'golden_formats.json'), 'w') as f:
        json.dump(gold, f)
    print('wrote golden_formats.json:', len(gold['pifpaf']), 'pifpaf cases,', len(gold['kitti']), 'kitti cases')


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'formats':
    golden_formats()


def golden_legacy():
    """The legacy MonolocoModel (architectures.py:105-176; 34 -> 256 -> 2) through the reference's own classes, on
    the pifpaf fixture with zero-centred inputs exactly as Loco.forward's 'monoloco' branch (net.py:95-100) does."""
    from monoloco.network.architectures import MonolocoModel
    from monoloco.network.process import unnormalize_bi
    torch.manual_seed(7)
    model = MonolocoModel(input_size=34, output_size=2, linear_size=256, p_dropout=0.2)
    with torch.no_grad():
        for name, mod in model.named_modules():
            if isinstance(mod, torch.nn.BatchNorm1d):  # non-trivial running statistics and affine parameters
                mod.running_mean.copy_(torch.randn_like(mod.running_mean) * 0.3)
                mod.running_var.copy_(torch.rand_like(mod.running_var) + 0.5)
                mod.weight.copy_(torch.rand_like(mod.weight) + 0.5)
                mod.bias.copy_(torch.randn_like(mod.bias) * 0.2)
        model.w2.bias.copy_(torch.tensor([12.0, -1.5]))  # d around 12 m, b/d around e^-1.5
    model.eval()
    anns = json.load(open(os.path.join(REF, 'tests', '002282.png.pifpaf.json')))
    im_size = (1238, 374)
    boxes, keypoints = preprocess_pifpaf(copy.deepcopy(anns), im_size, enlarge_boxes=False)
    kk = load_calibration('kitti', im_size)
    out = {}
    with torch.no_grad():
        kps = torch.tensor(keypoints)
        x = preprocess_monoloco(kps, torch.tensor(kk), zero_center=True)
        raw = model(x)
        out.update(kps=kps.numpy(), kk=np.array(kk, dtype=np.float64), x=x.numpy(), raw=raw.numpy(),
                   d=raw[:, 0:1].numpy(), bi=unnormalize_bi(raw).numpy(),
                   raw64=model.double()(x.double()).numpy())
    for k, v in np_sd(model.float().state_dict()).items():
        out['sd.' + k] = v
    np.savez_compressed(os.path.join(OUT, 'golden_legacy.npz'), **out)
    print('wrote golden_legacy.npz: raw range', raw.min().item(), raw.max().item())


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'legacy':
    golden_legacy()


def golden_mono_p():
    """Legacy 'monoloco_p': extract_outputs_mono (process.py:330-360) on seeded (m,9) outputs, and the reference's
    MonolocoModel(34 -> 256 -> 9) + extract_outputs_mono on the pifpaf fixture (Loco.forward branch net.py:102-104)."""
    from monoloco.network.architectures import MonolocoModel
    from monoloco.network.process import extract_outputs_mono
    torch.manual_seed(13)
    m = 300
    raw = torch.randn(m, 9)
    raw[:, 0] = raw[:, 0] * 6
    raw[:, 1] = raw[:, 1] * 1.5 + 1
    raw[:, 2] = torch.rand(m) * 40 + 1
    raw[::17, 2] = -raw[::17, 2]      # behind the camera: exercises the +-2pi wrap of the egocentric angle
    raw[:, 3] = torch.randn(m) * 0.5 - 2
    dic = extract_outputs_mono(raw.clone())
    out = {'raw': raw.numpy()}
    out.update(dic_to_np(dic, 'ex_'))
    model = MonolocoModel(input_size=34, output_size=9, linear_size=256, p_dropout=0.2)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn_like(mod.running_mean) * 0.3)
                mod.running_var.copy_(torch.rand_like(mod.running_var) + 0.5)
                mod.weight.copy_(torch.rand_like(mod.weight) + 0.5)
                mod.bias.copy_(torch.randn_like(mod.bias) * 0.2)
        model.w2.bias.copy_(torch.tensor([0.5, 1.0, 15.0, -1.5, 1.7, 0.6, 0.8, 0.3, 0.7]))
    model.eval()
    anns = json.load(open(os.path.join(REF, 'tests', '002282.png.pifpaf.json')))
    boxes, keypoints = preprocess_pifpaf(copy.deepcopy(anns), (1238, 374), enlarge_boxes=False)
    kk = load_calibration('kitti', (1238, 374))
    with torch.no_grad():
        x = preprocess_monoloco(torch.tensor(keypoints), torch.tensor(kk))
        net_raw = model(x)
        net_dic = extract_outputs_mono(net_raw)
    out.update(kps=np.array(keypoints, dtype=np.float32), kk=np.array(kk, dtype=np.float64), net_raw=net_raw.numpy())
    out.update(dic_to_np(net_dic, 'net_'))
    for k, v in np_sd(model.state_dict()).items():
        out['sd.' + k] = v
    np.savez_compressed(os.path.join(OUT, 'golden_mono_p.npz'), **out)
    print('wrote golden_mono_p.npz', sorted(k for k in out if not k.startswith('sd.')))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'mono_p':
    golden_mono_p()


def golden_activity():
    """Activity rules of monoloco/activity.py through the reference's own functions on seeded random scenes."""
    import monoloco.network  # noqa: F401  (resolves the reference's circular import)
    from monoloco.activity import check_f_formations, is_raising_hand, social_interactions
    rng = np.random.default_rng(17)
    gold = {'raising': [], 'fform': [], 'social_det': [], 'social_prob': []}
    for _ in range(400):
        xs = rng.uniform(0, 100, 17)
        ys = rng.uniform(0, 100, 17)
        if rng.random() < 0.6:    # plausible upper body so that all four answers occur
            sx = 50 + rng.uniform(-5, 5)
            xs[[5, 6]] = [sx + 12, sx - 12]
            ys[[5, 6]] = 40
            xs[[3, 4]] = [sx + 5, sx - 5]
            ys[0] = 25
            for hand, elbow, sh in ((9, 7, 5), (10, 8, 6)):
                xs[elbow] = xs[sh] + rng.uniform(-12, 12)
                ys[elbow] = ys[sh] + rng.uniform(-15, 20)
                xs[hand] = xs[elbow] + rng.uniform(-20, 20)
                ys[hand] = ys[elbow] + rng.uniform(-30, 20)
        kp = [xs.tolist(), ys.tolist(), rng.uniform(0, 1, 17).tolist()]
        gold['raising'].append({'kp': kp, 'out': is_raising_hand(kp)})
    for _ in range(400):
        n = int(rng.integers(2, 7))
        centers = (rng.uniform(-2, 2, (n, 2)) + [0, 6]).tolist()
        angles = rng.uniform(-math.pi, math.pi, n).tolist()
        radii = (0.3, 0.5, 1) if rng.random() < 0.5 else (0.3, 0.5)
        sd = bool(rng.random() < 0.5)
        i, j = rng.choice(n, 2, replace=False)
        gold['fform'].append({'centers': centers, 'angles': angles, 'radii': list(radii), 'sd': sd, 'i': int(i), 'j': int(j),
                              'out': bool(check_f_formations(int(i), int(j), centers, angles, radii, social_distance=sd))})
    for _ in range(150):
        n = int(rng.integers(1, 8))
        centers = (rng.uniform(-2.5, 2.5, (n, 2)) + [0, 7]).tolist()
        angles = rng.uniform(-math.pi, math.pi, n).tolist()
        dds = [math.hypot(c[0], c[1]) for c in centers]
        out = [bool(social_interactions(i, centers, angles, dds, stds=[0.1] * n, n_samples=1, threshold_dist=2.5,
                                        radii=(0.3, 0.5, 1))) for i in range(n)]
        gold['social_det'].append({'centers': centers, 'angles': angles, 'dds': dds, 'out': out})
    # probabilistic branch: two persons facing each other at 1 m (flag robustly on) and back to back (robustly off)
    for facing in (True, False):
        centers = [[-0.5, 6.0], [0.5, 6.0], [3.0, 12.0]]
        angles = ([0.0, math.pi, 1.0] if facing else [math.pi, 0.0, 1.0])
        dds = [math.hypot(c[0], c[1]) for c in centers]
        stds = [0.05, 0.05, 0.05]
        out = [bool(social_interactions(i, centers, angles, dds, stds=stds, n_samples=100, threshold_prob=0.25,
                                        threshold_dist=2.5, radii=(0.3, 0.5, 1))) for i in range(3)]
        gold['social_prob'].append({'centers': centers, 'angles': angles, 'dds': dds, 'stds': stds, 'out': out})
    with open(os.path.join(OUT, 'golden_activity.json'), 'w') as f:
        json.dump(gold, f)
    from collections import Counter
    print('raising', Counter(str(g['out']) for g in gold['raising']), 'fform', Counter(g['out'] for g in gold['fform']),
          'social', Counter(v for g in gold['social_det'] for v in g['out']), [g['out'] for g in gold['social_prob']])


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'activity':
    import math
    golden_activity()


def golden_wb1024():
    """Headline-width parity pins.

    * W-B at the reference's DEFAULT width: its own fixture training at hidden 1024 (run.py:101; the recipe of
      tests/test_train_mono.py / test_train_stereo.py: lr 0.001, -e 10 mono / -e 20 stereo, r_seed 1).  The two
      checkpoints are committed (ckpt_{mono,stereo}_h1024.npz, exact fp32) together with the reference's outputs on
      the 500 / 556 fixture poses, through Loco.forward and through the bare model in fp32 and fp64.
    * The hyper-parameter-search widths 512 and 2048 (train/hyp_tuning.py:52) and a width that is not a multiple of
      256 (the reference accepts any linear_size): seeded synthetic weights (regenerated on the GPU box by
      tests/synth.py, checksum-pinned here), reference fp32 + fp64 outputs stored.
    """
    tmp = '/tmp/make_golden'
    os.makedirs(tmp, exist_ok=True)
    torch.set_num_threads(8)
    g = {}
    kk = synth.KITTI_K
    dj = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-mono.json')))
    kps = torch.tensor(dj['train']['kps'] + dj['val']['kps'])[:, 0]        # (500, 3, 17)
    ds = json.load(open(os.path.join(REF, 'tests', 'sample_joints-kitti-stereo.json')))
    kps_s = torch.tensor(ds['train']['kps'] + ds['val']['kps'])[:, 0]     # (556, 3, 34)
    x_s = torch.tensor(ds['train']['X'] + ds['val']['X'])                 # (556, 68)
    kl, kr = kps_s[:, :, :17].contiguous(), kps_s[:, :, 17:].contiguous()
    conf = np.linspace(0.2, 1.0, len(kps)).astype(np.float32)

    sd_m = np_sd(train('mono', 1024, 10, tmp))
    sd_s = np_sd(train('stereo', 1024, 20, tmp))
    for name, sd in (('mono', sd_m), ('stereo', sd_s)):
        np.savez_compressed(os.path.join(OUT, 'ckpt_%s_h1024.npz' % name),
                            **{k: v for k, v in sd.items() if not k.endswith('num_batches_tracked')})
    with torch.no_grad():
        x = preprocess_monoloco(kps, torch.tensor(kk))
        net = Loco(model=ref_model(sd_m, 34, 9, 1024), mode='mono', linear_size=1024)
        dic = net.forward(kps.tolist(), kk)
        g['mono_raw'] = net.model(x).numpy()
        g['mono_raw64'] = ref_model(sd_m, 34, 9, 1024, torch.float64)(
            preprocess_monoloco(kps.double(), torch.tensor(kk, dtype=torch.float64))).numpy()
    g.update(dic_to_np(dic, 'mono_'))
    g['mono_xyz_pred'], g['mono_conf'] = geometry(kps, kk, dic['d'], dic['bi'], conf)
    with torch.no_grad():
        g['stereo_raw_fixture'] = ref_model(sd_s, 68, 10, 1024)(x_s).numpy()
        g['stereo_raw64_fixture'] = ref_model(sd_s, 68, 10, 1024, torch.float64)(x_s.double()).numpy()
        nl, nr = 40, 7
        net = Loco(model=ref_model(sd_s, 68, 10, 1024), mode='stereo', linear_size=1024)
        dic = net.forward(kl[:nl].tolist(), kk, keypoints_r=kr[:nr].tolist())
        inputs, _ = preprocess_monstereo(kl[:nl], kr[:nr], torch.tensor(kk))
        g['stereo_ava_raw_all'] = net.model(inputs).numpy()
    g.update(dic_to_np(dic, 'stereo_ava_'))
    g['stereo_ava_nl_nr'] = np.array([nl, nr])

    # other widths on seeded synthetic weights: (seed, in, out, hidden)
    for seed, in_f, out_f, hidden in ((21, 34, 9, 512), (22, 34, 9, 2048), (23, 34, 9, 600), (24, 68, 10, 200)):
        sd = synth.make_state_dict(seed, in_f, out_f, hidden)
        tag = 'w%d_' % hidden
        g[tag + 'checksum'] = np.float64(synth.checksum(sd))
        with torch.no_grad():
            xin = x if in_f == 34 else x_s
            g[tag + 'raw'] = ref_model(sd, in_f, out_f, hidden)(xin).numpy()
            g[tag + 'raw64'] = ref_model(sd, in_f, out_f, hidden, torch.float64)(xin.double()).numpy()
    np.savez_compressed(os.path.join(OUT, 'golden_wb1024.npz'), **g)
    d = g['mono_d']
    print('W-B-1024 mono: d %.2f..%.2f m, bi %.3f..%.3f, NaN z rows %d' %
          (d.min(), d.max(), g['mono_bi'].min(), g['mono_bi'].max(), int(np.isnan(g['mono_xyzd'][:, 2]).sum())))
    print('fp32-vs-fp64 noise of the reference: mono %.3e, stereo %.3e' %
          (np.abs(g['mono_raw'] - g['mono_raw64']).max(), np.abs(g['stereo_raw_fixture'] - g['stereo_raw64_fixture']).max()))
    for f in sorted(os.listdir(OUT)):
        if '1024' in f:
            print('  %-28s %8.1f KiB' % (f, os.path.getsize(os.path.join(OUT, f)) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'wb1024':
    golden_wb1024()


def golden_matching():
    """Ground-truth association on images with many boxes: the reference's get_iou_matches / reorder_matches / get_iou_matrix
    (monoloco/utils/iou.py:31-100) and the 'gt' / 'dds_real' / 'xyz_real' side of its post_process (net.py:170-190, 242-247) on
    seeded box sets (synth.make_boxes: 16, 256 and 2048 detections, with confidence ties, IoU ties, duplicated boxes).  The
    boxes are regenerated from the seed at test time; stored are the reference's results and the two np.argsort orders it saw
    (tie order of an unstable sort may depend on the CPU's SIMD dispatch: a test that meets a different order on its
    machine compares against the oracle, which calls the same np.argsort, and says so)."""
    from monoloco.utils.iou import get_iou_matches, get_iou_matches_matrix, get_iou_matrix, reorder_matches
    cases = []
    for m, g, seed, ties, iou_min in ((16, 16, 11, True, 0.3), (16, 5, 12, False, 0.3), (5, 40, 13, True, 0.5), (256, 256, 14, True, 0.3),
                                      (256, 300, 15, False, 0.3), (2048, 2048, 16, True, 0.3), (2048, 1500, 17, False, 0.45)):
        boxes, gt = synth.make_boxes(m, g, seed, ties=ties)
        matches = get_iou_matches(boxes, gt, iou_min=iou_min)
        ordered = reorder_matches(matches, boxes, mode='left_right')
        case = dict(m=m, g=g, seed=seed, ties=ties, iou_min=iou_min, matches=[list(p) for p in matches],
                    ordered=[list(p) for p in ordered],
                    argsort_conf=np.argsort([b[4] for b in boxes]).tolist(), argsort_left=np.argsort([b[0] for b in boxes]).tolist())
        if m <= 256:
            mat = get_iou_matrix(boxes, gt)
            case['iou_sum'] = float(mat.sum())
            case['iou_row3'] = mat[3].tolist()
            case['matches_matrix'] = [[int(a), int(b)] for a, b in get_iou_matches_matrix(boxes, gt, iou_min)]
        cases.append(case)
        print('matching %4d x %4d: %d matches' % (m, g, len(matches)))
    # post_process with ground truth at 256 persons: network outputs are stand-ins (post_process only reads d, bi, epi, yaw)
    m, g = 256, 200
    boxes, gt = synth.make_boxes(m, g, 21, ties=True)
    kps = synth.make_poses(m, 22)
    rng = np.random.default_rng(23)
    dic_in = {'d': torch.tensor(rng.uniform(2, 40, (m, 1)).astype(np.float32)), 'bi': torch.tensor(rng.uniform(0.1, 2, (m, 1)).astype(np.float32)),
              'epi': [0.] * m, 'yaw': (torch.tensor(rng.uniform(-3, 3, (m, 1)).astype(np.float32)),
                                       torch.tensor(rng.uniform(-3, 3, (m, 1)).astype(np.float32)))}
    dic_gt = {'boxes': gt, 'ys': [[0, 0, 0, 3.0 + 0.173 * j] for j in range(g)]}
    post = {}
    for reorder in (True, False):
        out = Loco.post_process(dic_in, copy.deepcopy(boxes), kps.tolist(), synth.KITTI_K, dic_gt=dic_gt, reorder=reorder)
        post['reorder_%d' % reorder] = {k: out[k] for k in ('gt', 'dds_real', 'xyz_real', 'boxes_gt', 'dds_pred', 'confs', 'xyz_pred',
                                                             'uv_centers', 'angles')}
        post['reorder_%d' % reorder]['boxes_x1'] = [b[0] for b in out['boxes']]
    json.dump({'cases': cases, 'post_256': post}, open(os.path.join(OUT, 'golden_matching.json'), 'w'))
    print('golden_matching.json %.1f KiB' % (os.path.getsize(os.path.join(OUT, 'golden_matching.json')) / 1024))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'matching':
    golden_matching()
