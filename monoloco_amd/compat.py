"""Drop the MI355X hot path into the reference's own module tree.

Two situations, one call (``monoloco_amd.compat.install()``):

* **The reference package is importable** (``import monoloco`` works -- a checkout or a pip install).  Nothing is
  replaced wholesale: the reference's modules keep every name this repository does not implement
  (``monoloco.utils.get_task_error``, ``open_image``, ``monoloco.prep``, ``monoloco.eval`` ...).  Only the names of
  the keypoint->3D path are re-bound -- on ``monoloco.network``, ``monoloco.network.net``,
  ``monoloco.network.process``, ``monoloco.network.architectures``, ``monoloco.utils``,
  ``monoloco.utils.camera`` and ``monoloco.utils.iou`` -- and in every already-imported ``monoloco.*`` module that holds a
  ``from ..network import Loco``-style copy of one of them (``monoloco.predict``, ``monoloco.eval.generate_kitti``,
  ``monoloco.visuals.printer`` ...).  Modules imported later pick the patched names up from the packages.
  The reference callers (predict.py:31, visuals/webcam.py:25, eval/generate_kitti.py:14-18,
  eval/eval_activity.py) then run unmodified on the HIP path.
* **Stand-alone** (no reference on the path): ``monoloco`` becomes an alias of this package, so
  ``from monoloco.network import Loco`` / ``from monoloco.utils import pixel_to_camera`` /
  ``from monoloco.train import Trainer`` resolve to the implementations here.  Only what this repository implements
  exists under that name.

``uninstall()`` restores the reference's own functions (used by the tests).
"""
import importlib
import importlib.util
import sys

# names of the path, by the reference module that defines them
_NET = ('Loco',)
_PROCESS = ('preprocess_pifpaf', 'prepare_pif_kps', 'factory_for_gt', 'load_calibration', 'preprocess_monoloco',
            'preprocess_monstereo', 'unnormalize_bi', 'extract_outputs', 'extract_outputs_mono', 'extract_labels',
            'extract_labels_aux', 'cluster_outputs', 'filter_outputs', 'laplace_sampling')
_ARCH = ('LocoModel', 'MonolocoModel')
_CAMERA = ('pixel_to_camera', 'get_keypoints', 'xyz_from_distance', 'to_cartesian', 'back_correct_angles')
# ground-truth association (utils/iou.py): the same matches, one native call per image instead of m x g Python calls (round 5) --
# Loco.post_process uses them here; the reference's other callers (eval/eval_kitti.py:21, eval/eval_activity.py:20,
# prep/preprocess_kitti.py:19) get them too
_IOU = ('calculate_iou', 'get_iou_matrix', 'get_iou_matches', 'get_iou_matches_matrix', 'reorder_matches')
_TRAIN = ('Trainer',)   # only with install(trainer=True): `python -m monoloco.run train` then trains on the HIP step

_saved = []  # (module, name, original object, existed) of everything install() re-bound
_pairs = []  # (name, our object, the reference's object)


def _reference_available():
    """True if a real reference package (not an alias made by this module) is imported or importable."""
    mod = sys.modules.get('monoloco')
    if mod is not None:
        return getattr(mod, '__name__', '') == 'monoloco' and not getattr(mod, '_monoloco_amd_alias', False)
    try:
        spec = importlib.util.find_spec('monoloco')
    except (ImportError, ValueError):
        spec = None
    return spec is not None


def _bind(mod, name, obj):
    if getattr(mod, name, None) is obj:
        return
    _saved.append((mod, name, getattr(mod, name, None), hasattr(mod, name)))
    setattr(mod, name, obj)


def install(trainer=False):
    """Re-bind the keypoint->3D path of ``monoloco`` to the MI355X implementations; returns the ``monoloco`` package.
    trainer=True also re-binds ``monoloco.train.Trainer`` (run.py:153-170, train/hyp_tuning.py) to the HIP training step --
    opt-in, because the inference path is what a caller asks for by default and the reference's Trainer also runs on CPU."""
    import importlib.util  # noqa: F401  (find_spec)
    from . import activity, formats, network, train, utils  # noqa: F401  (import everything that gets an alias)
    from .network import architectures, net, process
    from .utils import camera, iou

    if not _reference_available():
        # stand-alone: `monoloco` IS this package; every loaded submodule gets the matching alias so that
        # `import monoloco.network.process` finds the module object that already exists (no second copy)
        pkg = sys.modules[__package__]
        pkg._monoloco_amd_alias = True
        for name, mod in list(sys.modules.items()):
            if name == __package__ or name.startswith(__package__ + '.'):
                sys.modules['monoloco' + name[len(__package__):]] = mod
        return pkg

    import monoloco  # the reference
    ref = {key: importlib.import_module('monoloco.' + key)
           for key in ('network', 'network.net', 'network.process', 'network.architectures', 'utils', 'utils.camera', 'utils.iou')}
    groups = [(_NET, net, 'network.net'), (_PROCESS, process, 'network.process'), (_ARCH, architectures, 'network.architectures'),
              (_CAMERA, camera, 'utils.camera'), (_IOU, iou, 'utils.iou')]
    if trainer:
        from .train import trainer as our_trainer
        for key in ('train', 'train.trainer'):
            ref[key] = importlib.import_module('monoloco.' + key)
        groups.append((_TRAIN, our_trainer, 'train.trainer'))
    new = {}
    for names, src, _ in groups:
        for name in names:
            if hasattr(src, name):
                new[name] = getattr(src, name)
    originals = {}
    for names, _, key in groups:
        for name in names:
            if name in new and hasattr(ref[key], name):
                originals[name] = getattr(ref[key], name)
    # the defining modules and the packages that re-export (network/__init__.py:2-4, utils/__init__.py:8-9)
    for name, obj in new.items():
        for key, mod in ref.items():
            if hasattr(mod, name):
                _bind(mod, name, obj)
    # copies taken by `from ..network import Loco` in modules that are already imported
    for modname, mod in list(sys.modules.items()):
        if mod is None or not (modname == 'monoloco' or modname.startswith('monoloco.')):
            continue
        for name, orig in originals.items():
            if getattr(mod, name, None) is orig:
                _bind(mod, name, new[name])
    _pairs[:] = [(name, new[name], orig) for name, orig in originals.items()]
    return monoloco


def uninstall():
    """Undo install(): restore the reference's own objects / remove the stand-alone aliases."""
    while _saved:
        mod, name, orig, had = _saved.pop()
        if had:
            setattr(mod, name, orig)
        else:
            delattr(mod, name)
    # modules imported AFTER install() copied our objects straight from the patched packages
    for modname, mod in list(sys.modules.items()):
        if mod is not None and modname.startswith('monoloco.'):
            for name, ours, orig in _pairs:
                if getattr(mod, name, None) is ours:
                    setattr(mod, name, orig)
    del _pairs[:]
    pkg = sys.modules.get(__package__)
    if pkg is not None and getattr(pkg, '_monoloco_amd_alias', False):
        for name in [n for n in sys.modules if n == 'monoloco' or n.startswith('monoloco.')]:
            del sys.modules[name]
        pkg._monoloco_amd_alias = False
