"""monoloco_amd.compat: the reference's callers keep importing `monoloco.network` / `monoloco.utils` and get the
MI355X path.  Each scenario runs in its own interpreter so that the module
re-binding never leaks into the test process."""
import copy
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
G = os.path.join(ROOT, 'tests', 'golden')

_STUBS = r'''
import os, sys, types
sys.dont_write_bytecode = True
class _Any:
    def __init__(self, *a, **k): pass
    def __getattr__(self, n): return _Any()
    def __call__(self, *a, **k): return _Any()
def stub(name, **attrs):
    m = types.ModuleType(name); m.__dict__.update(attrs); sys.modules[name] = m; return m
tv = stub('torchvision'); tv.transforms = stub('torchvision.transforms'); tv.models = stub('torchvision.models')
op = stub('openpifpaf', Predictor=_Any)
for sub in ('datasets', 'decoder', 'network', 'visualizer', 'show', 'logger', 'predict'):
    setattr(op, sub, stub('openpifpaf.' + sub))
op.predict.out_name = lambda *a, **k: None
'''


def _run(code, cwd):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', code], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'monoloco')), reason="needs the reference checkout (build container only)")
def test_install_patches_the_reference_in_place(tmp_path, hip_lib):
    """With the reference importable: its callers import fine (predict.py:30-32, eval/generate_kitti.py:14-21,
    visuals/printer.py), hold OUR Loco / preprocess_pifpaf / pixel_to_camera -- also the copies taken before
    install() -- while everything outside the path (get_task_error, open_image, ...) is still the reference's."""
    os.makedirs(tmp_path / 'data' / 'logs')  # monoloco.eval asserts it relative to the cwd
    code = _STUBS + r'''
sys.path.insert(0, %r); sys.path.insert(0, %r)
import monoloco.predict                      # imported BEFORE install(): holds `from .network import Loco` copies
ref_loco = monoloco.predict.Loco
import monoloco_amd.compat as C
pkg = C.install()
import monoloco.visuals.printer, monoloco.eval.generate_kitti, monoloco.visuals.webcam
import monoloco, monoloco_amd.network as N, monoloco_amd.utils as U
assert pkg is monoloco and monoloco.__file__.startswith(%r)
assert monoloco.network.Loco is N.Loco and monoloco.network.net.Loco is N.Loco
assert monoloco.predict.Loco is N.Loco and monoloco.eval.generate_kitti.Loco is N.Loco and monoloco.visuals.webcam.Loco is N.Loco
assert monoloco.predict.preprocess_pifpaf is N.preprocess_pifpaf and monoloco.predict.factory_for_gt is N.factory_for_gt
assert monoloco.predict.load_calibration is N.load_calibration
assert monoloco.network.process.extract_outputs is N.extract_outputs
assert monoloco.eval.generate_kitti.pixel_to_camera is U.pixel_to_camera
assert monoloco.eval.generate_kitti.xyz_from_distance is U.xyz_from_distance
assert monoloco.visuals.printer.pixel_to_camera is U.pixel_to_camera
assert monoloco.utils.camera.get_keypoints is U.get_keypoints and monoloco.utils.back_correct_angles is U.back_correct_angles
import monoloco_amd.utils.iou as UI
assert monoloco.utils.get_iou_matches is UI.get_iou_matches and monoloco.utils.iou.reorder_matches is UI.reorder_matches
# outside the path: untouched reference objects
for name in ('get_task_error', 'make_new_directory', 'factory_basename', 'get_category', 'split_training',
             'get_calibration', 'open_image', 'project_3d'):
    assert getattr(monoloco.utils, name).__module__.startswith('monoloco.utils.'), name
assert monoloco.eval.generate_kitti.factory_basename.__module__ == 'monoloco.utils.kitti'
# host functions of the path behave (no GPU needed): the reference fixture through the patched module path
import json, copy
ann = json.load(open(%r))
boxes, kps = monoloco.network.preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
assert len(boxes) == 16 and len(kps[0]) == 3 and len(kps[0][0]) == 17
# the reference's Trainer stays its own unless asked for
import monoloco.train, monoloco.train.hyp_tuning, monoloco_amd.train as T
ref_trainer = monoloco.train.Trainer
assert ref_trainer.__module__ == 'monoloco.train.trainer'
C.uninstall()
assert monoloco.network.Loco is ref_loco and monoloco.predict.Loco is ref_loco
assert monoloco.eval.generate_kitti.pixel_to_camera.__module__ == 'monoloco.utils.camera'
# install(trainer=True): run.py:153-170 and hyp_tuning.py then construct the HIP Trainer
C.install(trainer=True)
assert monoloco.train.Trainer is T.Trainer and monoloco.train.trainer.Trainer is T.Trainer
assert monoloco.train.hyp_tuning.Trainer is T.Trainer and monoloco.network.Loco is N.Loco
C.uninstall()
assert monoloco.train.Trainer is ref_trainer and monoloco.train.hyp_tuning.Trainer is ref_trainer
print('ok')
''' % (REF, ROOT, REF, os.path.join(G, 'pifpaf_002282.json'))
    assert _run(code, str(tmp_path)).strip().endswith('ok')


def test_standalone_alias(tmp_path, hip_lib):
    """Without the reference on the path `monoloco` is an alias of this package: the reference's import lines work,
    there is exactly one copy of every module, and names outside the implemented path fail loudly."""
    code = r'''
import sys
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
assert not any(p.rstrip('/') == '/root/reference' for p in sys.path)
import monoloco_amd.compat as C
C.install()
from monoloco.network import Loco, factory_for_gt, load_calibration, preprocess_pifpaf          # predict.py:31
from monoloco.network.process import preprocess_pifpaf as pp2, preprocess_monoloco, extract_outputs  # generate_kitti.py:15
from monoloco.utils import get_keypoints, pixel_to_camera, xyz_from_distance, get_iou_matches   # net.py:13, generate_kitti.py:17
from monoloco.train import Trainer
from monoloco.network.architectures import LocoModel, MonolocoModel
import monoloco, monoloco.network.net, monoloco_amd.network.net
assert monoloco.network.net is monoloco_amd.network.net and pp2 is preprocess_pifpaf
assert Loco is monoloco_amd.network.Loco and Loco.LINEAR_SIZE_MONO == 256 and Loco.N_SAMPLES == 100
assert load_calibration('kitti', (1242, 375))[0][0] > 700
try:
    from monoloco.utils import get_task_error
except ImportError:
    pass
else:
    raise AssertionError('a name outside the path must not appear out of nowhere')
C.uninstall()
assert 'monoloco' not in sys.modules and 'monoloco.network' not in sys.modules
print('ok')
''' % ROOT
    assert _run(code, str(tmp_path)).strip().endswith('ok')


@pytest.mark.gpu
def test_generate_kitti_call_sequence_through_compat(hip_lib, cuda_device):
    """The GenerateKitti inner loop (eval/generate_kitti.py:41-48, 114-132) written with the reference's import lines,
    on the reference's pifpaf fixture with the reference-trained fixture weights; checked against the goldens recorded
    from the real Loco.forward, then through the KITTI txt writer."""
    import monoloco_amd.compat as C
    C.install()
    try:
        from monoloco.network import Loco
        from monoloco.network.process import preprocess_pifpaf
        from monoloco.utils import get_keypoints, pixel_to_camera, xyz_from_distance
        from monoloco.network.architectures import LocoModel
        import monoloco_amd
        from monoloco_amd import formats
        assert Loco is monoloco_amd.network.Loco
        sd = {k: torch.tensor(v) for k, v in np.load(os.path.join(G, 'ckpt_mono_h256.npz')).items()}
        model = LocoModel(34, 9, 256)
        model.load_state_dict(sd)
        net = Loco(model=model, mode='mono', device=cuda_device, n_dropout=0, p_dropout=0.2, linear_size=256)
        ann = json.load(open(os.path.join(G, 'pifpaf_002282.json')))
        cj = json.load(open(os.path.join(G, 'golden_c1.json')))
        cn = np.load(os.path.join(G, 'golden_c1.npz'))
        kk = cj['calib']['kitti_1238_374']
        boxes, keypoints = preprocess_pifpaf(copy.deepcopy(ann), im_size=(1238, 374), enlarge_boxes=False)
        assert boxes == cj['pre_predict']['boxes'] and keypoints == cj['pre_predict']['keypoints']
        dic_out = net.forward(keypoints, kk)
        all_outputs = [dic_out['xyzd'], dic_out['bi'], dic_out['epi'], dic_out['yaw'], dic_out['h'], dic_out['w'], dic_out['l']]
        zzs = [float(el[2]) for el in dic_out['xyzd']]
        assert len(zzs) == len(boxes) == 16
        assert np.abs(dic_out['xyzd'].numpy()[:, [0, 1, 3]] - cn['fwd_B_xyzd'][:, [0, 1, 3]]).max() <= 1e-4
        assert np.abs(dic_out['bi'].numpy() - cn['fwd_B_bi']).max() <= 1e-4
        # the geometric side calls of the same loop (generate_kitti.py:17, geom_baseline): HIP-backed utils
        uv_c = get_keypoints(keypoints, mode='center')
        xyz = xyz_from_distance(dic_out['d'], pixel_to_camera(uv_c, kk, 1))
        assert np.abs(xyz.numpy() - np.array(cj['post_B']['xyz_pred'])).max() <= 1e-4
        path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'compat_002282.txt')
        formats.save_txts(path, copy.deepcopy(boxes), all_outputs, [kk, None], net='monoloco_pp',
                          cat=[0.0] * 15 + [1.0])        # get_category's per-person cyclist score (generate_kitti.py:110)
        lines = open(path).read().strip().split('\n')
        assert len(lines) == 16 and all(len(ln.split()) == 18 for ln in lines)
        assert [ln.split()[0] for ln in lines] == ['Pedestrian'] * 15 + ['Cyclist']
    finally:
        C.uninstall()
