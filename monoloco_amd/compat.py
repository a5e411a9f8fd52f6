"""Make ``import monoloco.network`` / ``monoloco.utils`` resolve to this package.

The reference's callers (monoloco/predict.py:31, visuals/webcam.py:25, eval/generate_kitti.py:14-15)
import ``Loco``, ``preprocess_pifpaf`` ... from ``monoloco.network``.  ``install()`` registers the
MI355X implementations under those module paths so such callers run unmodified:

    import monoloco_amd.compat; monoloco_amd.compat.install()
"""
import sys
import types


def install(force=False):
    from . import network, utils
    from .network import architectures, net, process
    if 'monoloco' in sys.modules and not force:
        pkg = sys.modules['monoloco']
    else:
        pkg = types.ModuleType('monoloco')
        pkg.__path__ = []
        sys.modules['monoloco'] = pkg
    for name, mod in (('network', network), ('utils', utils), ('network.net', net), ('network.process', process),
                      ('network.architectures', architectures)):
        sys.modules['monoloco.' + name] = mod
    pkg.network = network
    pkg.utils = utils
    return pkg
