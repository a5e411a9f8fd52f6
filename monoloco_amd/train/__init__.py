"""Host mirror of ``monoloco.train`` for the training step (reference monoloco/train/__init__.py)."""
from .hip_trainer import HipTrainer
from .trainer import KeypointsDataset, Trainer

__all__ = ['HipTrainer', 'Trainer', 'KeypointsDataset']
