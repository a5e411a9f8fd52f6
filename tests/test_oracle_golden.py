"""Pin the CPU oracle against golden vectors produced by the REAL reference (oracle/make_golden.py)
and against the data pins held by the reference's own fixtures.  Runs on CPU."""
import json
import os

import numpy as np
import pytest
import torch

import synth
from oracle import monoloco_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def gold():
    return dict(np.load(os.path.join(G, 'golden_path.npz')))


def _sd(name):
    return {k: torch.tensor(v) for k, v in np.load(os.path.join(G, name)).items()}


def _weights(tag, mode):
    if tag == 'A':
        sd = synth.make_state_dict(1, 34, 9, 1024) if mode == 'mono' else synth.make_state_dict(3, 68, 10, 1024)
        return {k: torch.tensor(v) for k, v in sd.items()}
    return _sd('ckpt_%s_h256.npz' % mode)


def test_synth_weights_reproduce(gold):
    """The seeded weight generator must give the very weights the goldens were made with."""
    assert synth.checksum(synth.make_state_dict(1, 34, 9, 1024)) == float(gold['synth_checksum_mono'])
    assert synth.checksum(synth.make_state_dict(3, 68, 10, 1024)) == float(gold['synth_checksum_stereo'])


def test_preprocess_data_pin_mono(gold):
    """reference fixture: stored X rows are bit-exactly preprocess_monoloco(kps, K) for one of the file's Ks."""
    kps = torch.tensor(gold['mono_kps'])
    for i, k in enumerate(gold['mono_unique_k']):
        rows = gold['mono_k_index'] == i
        x = O.preprocess_monoloco(kps[rows], torch.tensor(k))
        assert torch.equal(x, torch.tensor(gold['mono_x_fixture'][rows]))


def test_preprocess_data_pin_stereo(gold):
    kl, kr = torch.tensor(gold['stereo_kps_l']), torch.tensor(gold['stereo_kps_r'])
    for i, k in enumerate(gold['stereo_unique_k']):
        rows = gold['stereo_k_index'] == i
        xl = O.preprocess_monoloco(kl[rows], torch.tensor(k))
        xr = O.preprocess_monoloco(kr[rows], torch.tensor(k))
        assert torch.equal(torch.cat((xl, xl - xr), 1), torch.tensor(gold['stereo_x_fixture'][rows]))


def test_pixel_to_camera_linearity():
    """The reference's own unit test of this path (tests/test_utils.py:18-25), exact equality."""
    a = O.pixel_to_camera([[1000., 400.]], synth.KITTI_K, 1)[0] * 10
    b = O.pixel_to_camera([[1000., 400.]], synth.KITTI_K, 10)[0]
    assert torch.equal(a, b)


@pytest.mark.parametrize("tag", ["A", "B"])
def test_mono_path_matches_reference(gold, tag):
    sd = _weights(tag, 'mono')
    kps = torch.tensor(gold['mono_kps'])
    out = O.forward_mono(sd, kps, synth.KITTI_K, box_conf=gold['mono_conf'])
    p = 'mono_%s_' % tag
    if tag == 'A':
        assert torch.equal(out['inputs'], torch.tensor(gold['mono_x_kitti']))
    assert np.abs(out['raw'].numpy() - gold[p + 'raw']).max() <= 5e-6   # sgemm blocking / thread count only
    raw = torch.tensor(gold[p + 'raw'])
    ext = O.extract_outputs(raw)                                         # same inputs -> same bits
    for key in ('h', 'w', 'l', 'ori', 'bi', 'xyzd', 'd'):
        a, b = ext[key].numpy(), gold[p + key]
        assert np.array_equal(a, b, equal_nan=True), key
    assert np.array_equal(ext['yaw'][0].numpy(), gold[p + 'yaw_pred'])
    assert np.array_equal(ext["yaw"][1].numpy(), gold[p + "yaw_ego"], equal_nan=True)
    xyz, conf = O.back_project(kps, synth.KITTI_K, ext['d'], ext['bi'], gold['mono_conf'])
    assert np.array_equal(xyz.numpy(), gold[p + 'xyz_pred'])
    assert np.abs(conf.numpy() / gold[p + 'conf'] - 1).max() <= 1e-6
    # fp64 mode of the oracle against the reference run in fp64
    out64 = O.forward_mono(sd, kps, synth.KITTI_K, dtype=torch.float64)
    assert np.abs(out64['raw'].numpy() - gold[p + 'raw64']).max() <= 1e-10


@pytest.mark.parametrize("tag", ["A", "B"])
def test_stereo_path_matches_reference(gold, tag):
    sd = _weights(tag, 'stereo')
    p = 'stereo_%s_' % tag
    raw = O.loco_forward(sd, torch.tensor(gold['stereo_x_fixture']))
    assert np.abs(raw.numpy() - gold[p + 'raw_fixture']).max() <= 5e-6
    nl, nr = gold['stereo_ava_nl_nr']
    kl, kr = torch.tensor(gold['stereo_kps_l'][:nl]), torch.tensor(gold['stereo_kps_r'][:nr])
    out = O.forward_stereo(sd, kl, kr, synth.KITTI_K)
    if tag == 'A':
        assert torch.equal(out['inputs'], torch.tensor(gold['stereo_ava_inputs']))
    assert np.abs(out['raw_all'].numpy() - gold[p + 'ava_raw_all']).max() <= 5e-6
    ref_raw_all = torch.tensor(gold[p + 'ava_raw_all'])
    sel, _ = O.cluster_and_filter(ref_raw_all, int(nr))
    ext = O.extract_outputs(sel)
    for key in ('h', 'w', 'l', 'ori', 'bi', 'xyzd', 'd', 'aux'):
        assert np.array_equal(ext[key].numpy(), gold[p + 'ava_' + key], equal_nan=True), key
    if tag == 'A':
        out0 = O.forward_stereo(sd, kl[:5], None, synth.KITTI_K)
        assert np.abs(out0['d'].numpy() - gold['stereo_A_noright_d']).max() <= 5e-6


def test_filter_keeps_ties():
    """filter_outputs keeps every row that ties for the maximum (reference process.py:325-326)."""
    o = torch.zeros((2 * 3, 10))
    o[:, -1] = torch.tensor([0.1, 0.7, 0.7, 0.3, 0.2, 0.1])
    o[:, 2] = torch.arange(6.)
    sel, mask = O.cluster_and_filter(o, 3)
    assert sel.shape[0] == 3 and mask.tolist() == [[False, True, True], [True, False, False]]
    assert sel[:, 2].tolist() == [1., 2., 3.]


def test_oracle_z_nan_semantics():
    raw = torch.tensor([[0.3, 0.1, 5.0, -1., 0, 0, 0, 0.2, 0.3]])
    ext = O.extract_outputs(raw)
    # sin(0.1)^2 cos(0.3)^2 + cos(0.1)^2 <= 1, so z is finite here; force the NaN branch
    raw2 = raw.clone()
    raw2[0, 2] = -5.0  # negative d: x^2+y^2 <= d^2 still holds, z = sqrt(...) finite; bi negative
    assert torch.isfinite(ext['xyzd']).all()
    assert O.extract_outputs(raw2)['bi'][0, 0] < 0


def test_legacy_monoloco_model_oracle_vs_reference():
    """Legacy MonolocoModel (34 -> 256 -> 2) and the 'monoloco' branch of Loco.forward (net.py:95-100) against the
    reference's own classes on the pifpaf fixture (oracle/make_golden.py legacy)."""
    g = dict(np.load(os.path.join(G, 'golden_legacy.npz')))
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd.')}
    res = O.forward_legacy_monoloco(sd, torch.tensor(g['kps']), g['kk'].tolist())
    assert torch.equal(res['x'], torch.tensor(g['x']))          # zero-centred inputs: bit exact
    assert (res['raw'] - torch.tensor(g['raw'])).abs().max().item() <= 5e-6
    assert (res['d'] - torch.tensor(g['d'])).abs().max().item() <= 5e-6
    assert (res['bi'] - torch.tensor(g['bi'])).abs().max().item() <= 5e-6
    raw64 = O.monoloco_forward(sd, torch.tensor(g['x']), dtype=torch.float64)
    assert (raw64 - torch.tensor(g['raw64'])).abs().max().item() <= 1e-12


def test_extract_outputs_mono_oracle_vs_reference():
    """Legacy 'monoloco_p' post-processing (process.py:330-360) against the reference's own function."""
    g = dict(np.load(os.path.join(G, 'golden_mono_p.npz')))
    dic = O.extract_outputs_mono(torch.tensor(g['raw']))
    for key in ('xyz', 'zb', 'h', 'w', 'l', 'ori', 'xyzd', 'd', 'bi'):
        assert torch.equal(dic[key], torch.tensor(g['ex_' + key])), key
    assert torch.equal(dic['yaw'][0], torch.tensor(g['ex_yaw_pred']))
    assert (dic['yaw'][1] - torch.tensor(g['ex_yaw_ego'])).abs().max().item() <= 5e-7
    assert (torch.tensor(g['ex_yaw_ego']).abs() <= np.pi + 1e-6).all()
    sd = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith('sd.')}
    x = O.preprocess_monoloco(torch.tensor(g['kps']), g['kk'].tolist())
    raw = O.monoloco_forward(sd, x)
    assert (raw - torch.tensor(g['net_raw'])).abs().max().item() <= 5e-6
