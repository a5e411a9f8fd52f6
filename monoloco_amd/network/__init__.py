"""Host mirror of ``monoloco.network`` (reference monoloco/network/__init__.py:2-4)."""
from .net import Loco
from .process import extract_labels, extract_labels_aux, extract_outputs, factory_for_gt, load_calibration, \
    preprocess_pifpaf, unnormalize_bi  # noqa: F401  (same re-exports as the reference)

__all__ = ['Loco', 'load_calibration', 'factory_for_gt', 'preprocess_pifpaf', 'unnormalize_bi', 'extract_outputs',
           'extract_labels', 'extract_labels_aux']
