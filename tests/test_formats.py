"""On-disk formats either side of the path: the oracle restatement and the native (host-only C-ABI) reader /
writer against goldens produced by the real reference (oracle/make_golden.py formats).  Bit-exact: doubles
compare with ==, texts compare as strings.  Nothing here needs a GPU (ml_pifpaf_* / ml_kitti_txt_format make
no HIP calls)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import formats_oracle as FO

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, 'golden', 'golden_formats.json')))
TEXTS = dict(GOLD['texts'], fixture=open(os.path.join(HERE, 'golden', 'pifpaf_002282.json')).read())


def _kw(case):
    kw = dict(case['kwargs'])
    if 'im_size' in kw:
        kw['im_size'] = tuple(kw['im_size'])
    return kw


def _unplain(x):
    if isinstance(x, dict) and 'tensor' in x:
        return torch.tensor(x['tensor'])
    if isinstance(x, dict) and 'tuple' in x:
        return tuple(_unplain(v) for v in x['tuple'])
    return x


@pytest.mark.parametrize('case', GOLD['pifpaf'], ids=lambda c: '%s-%s' % (c['text'], json.dumps(c['kwargs'])))
def test_oracle_preprocess_pifpaf_matches_reference(case):
    boxes, kps = FO.read_pifpaf_text(TEXTS[case['text']], **_kw(case))
    assert boxes == case['boxes']
    assert kps == case['keypoints']


@pytest.mark.parametrize('case', GOLD['kitti'], ids=lambda c: c['net'])
def test_oracle_kitti_txt_matches_reference(case):
    assert _oracle_kitti(case) == case['text']


def _oracle_kitti(case):
    outs = [_unplain(o) for o in case['outputs']]
    net = case['net']
    if net in ('monoloco_pp', 'monstereo'):
        xyzd, bis, epis, yaws, hs, ws, ls = outs
        hwls = [[float(hs[i]), float(ws[i]), float(ls[i])] for i in range(len(case['boxes']))]
        return FO.kitti_txt(case['boxes'], xyzd[:, 0:3], bis, epis, yaws[0], yaws[1], hwls, cat=case['cat'],
                            conf_scale=0.035 if net == 'monoloco_pp' else 0.033)
    if net in ('monoloco', 'geometric'):
        dds, bis, epis, zzs_geom, xy_centers = outs
        from oracle import monoloco_oracle as O
        xyz = O.xyz_from_distance(dds, xy_centers)
        return FO.kitti_txt(case['boxes'], xyz, bis, epis, zzs_geom=zzs_geom if net == 'geometric' else None,
                            cat=case['cat'], conf_scale=0.05)
    xyz, bis, epis, _, _ = outs
    return FO.kitti_txt(case['boxes'], xyz, bis, epis, tt=case['params'][1], cat=case['cat'], conf_scale=0.05)


# ---------------------------------------------------------------- native reader / writer
@pytest.mark.parametrize('case', GOLD['pifpaf'], ids=lambda c: '%s-%s' % (c['text'], json.dumps(c['kwargs'])))
def test_native_pifpaf_reader_bit_exact(hip_lib, case):
    from monoloco_amd import formats
    boxes, kps = formats.parse_pifpaf_text(TEXTS[case['text']], **_kw(case))
    assert boxes.shape == (len(case['boxes']), 5) and kps.shape == (len(case['boxes']), 3, 17)
    assert boxes.tolist() == [[float(v) for v in b] for b in case['boxes']]
    assert kps.tolist() == [[[float(v) for v in row] for row in k] for k in case['keypoints']]


def test_native_pifpaf_reader_file_api(hip_lib, tmp_path):
    from monoloco_amd import formats
    from monoloco_amd.network import preprocess_pifpaf
    path = os.path.join(HERE, 'golden', 'pifpaf_002282.json')
    boxes, kps = formats.read_pifpaf_json(path, im_size=(1238, 374), enlarge_boxes=False)
    b_py, k_py = preprocess_pifpaf(json.load(open(path)), im_size=(1238, 374), enlarge_boxes=False)
    assert kps.dtype == torch.float32 and torch.equal(kps, torch.tensor(k_py))
    assert boxes.tolist() == b_py
    b_l, k_l = formats.load_pifpaf(path, im_size=(1238, 374), enlarge_boxes=False)
    assert b_l == b_py and k_l == k_py


@pytest.mark.parametrize('text,code', [
    ('{"keypoints": []}', 'top-level array'),
    ('[{"keypoints": [1, 2, 3], "bbox": [0, 0, 1, 1]}]', "expected 51"),
    ('[{"bbox": [0, 0, 1, 1]}]', "no 'keypoints'"),
    ('[{"keypoints": [%s]}]' % ', '.join(['1'] * 51), "no 'bbox'"),
    ('[{"keypoints": [%s], "bbox": [0, 0, 1]}]' % ', '.join(['1'] * 51), 'expected 4'),
    ('[{"keypoints": [%s], "bbox": [100, 100, 0, 0]}]' % ', '.join(['1'] * 51), 'Bounding box <=0'),
    ('[{"keypoints": [%s], "bbox": [0, 0, 1, 1]},]' % ', '.join(['1'] * 51), 'syntax error'),
    ('[{"keypoints": [%s], "bbox": [0, 0, 1, 1]}] x' % ', '.join(['1'] * 51), 'trailing data'),
    ('[{"keypoints": [1, 2', 'syntax error'),
])
def test_native_pifpaf_reader_errors(hip_lib, text, code):
    from monoloco_amd import _lib, formats
    with pytest.raises(_lib.MonolocoHipError) as exc:
        formats.parse_pifpaf_text(text)
    assert code in str(exc.value)


def test_native_pifpaf_reader_number_forms(hip_lib):
    """Exponents, negative zero, ints, whitespace, duplicate keys (the last one wins, like a dict)."""
    from monoloco_amd import formats
    kp = ['1e2', '-0.0', '2.5E-1'] * 17
    text = ' [ {"bbox" : [9,9,9,9], "keypoints":[%s] ,\n "bbox":[1.5e1, 2, 30, 4.25e+1]} ]\n' % ' , '.join(kp)
    boxes, kps = formats.parse_pifpaf_text(text)
    b_ref, k_ref = FO.read_pifpaf_text(text)
    assert boxes.tolist() == b_ref and kps.tolist() == k_ref
    assert np.signbit(kps[0, 1, 0])


def _save_and_compare(case, tmp_path):
    from monoloco_amd import formats
    outs = [_unplain(o) for o in case['outputs']]
    path = str(tmp_path / 'out.txt')
    formats.save_txts(path, [list(b) for b in case['boxes']], outs, case['params'], net=case['net'], cat=case['cat'])
    assert open(path).read() == case['text']


@pytest.mark.parametrize('case', [c for c in GOLD['kitti'] if c['net'] in ('monoloco_pp', 'monstereo', 'baseline')],
                         ids=lambda c: c['net'])
def test_native_save_txts_matches_reference(hip_lib, tmp_path, case):
    _save_and_compare(case, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [c for c in GOLD['kitti'] if c['net'] in ('monoloco', 'geometric')],
                         ids=lambda c: c['net'])
def test_native_save_txts_legacy_nets_use_hip_xyz_from_distance(hip_lib, cuda_device, tmp_path, case):
    """'monoloco' / 'geometric' back-project with xyz_from_distance first (generate_kitti.py:213-214) -- here the
    HIP kernel; '%f' keeps 6 decimals, so the text is compared number by number at 2e-6."""
    from monoloco_amd import formats
    outs = [_unplain(o) for o in case['outputs']]
    path = str(tmp_path / 'out.txt')
    formats.save_txts(path, [list(b) for b in case['boxes']], outs, case['params'], net=case['net'], cat=case['cat'])
    got, ref = open(path).read().split('\n'), case['text'].split('\n')
    assert len(got) == len(ref)
    for lg, lr in zip(got, ref):
        tg, tr = lg.split(), lr.split()
        assert tg[:3] == tr[:3] and len(tg) == len(tr)
        for a, b in zip(tg[3:], tr[3:]):
            assert abs(float(a) - float(b)) <= 2e-6 * max(1.0, abs(float(b)))


def test_native_kitti_txt_nan_and_empty(hip_lib):
    from monoloco_amd import formats
    assert formats.kitti_txt([], np.zeros((0, 3)), [], [], cat=[]) == ""
    boxes = [[1., 2., 3., 4., 0.5]]
    xyz = np.array([[1., 2., float('nan')]])
    txt = formats.kitti_txt(boxes, xyz, [0.5], [0.], cat=[0.])
    assert txt == FO.kitti_txt(boxes, xyz, [0.5], [0.], cat=[0.], conf_scale=0.035)
    assert '-nan' not in txt and 'nan' in txt


def test_write_monoloco_json_roundtrip(tmp_path):
    from monoloco_amd import formats
    dic = {'xyz_pred': [[1.0, 2.5, 3.25]], 'boxes': [[0.1, 0.2, 0.3, 0.4, 0.5]], 'gt': [True]}
    path = formats.write_monoloco_json(str(tmp_path / 'img.png'), dic)
    assert path.endswith('img.png.monoloco.json') and json.load(open(path)) == dic
