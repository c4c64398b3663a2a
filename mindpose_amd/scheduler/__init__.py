"""``mindpose.scheduler`` surface: ``create_lr_scheduler`` and the two warm-up schedules (mindpose/scheduler/__init__.py)."""
from ..utils.lr import WarmupCosineDecayLR, WarmupMultiStepDecayLR, create_lr_scheduler  # noqa: F401

__all__ = ["create_lr_scheduler", "WarmupCosineDecayLR", "WarmupMultiStepDecayLR"]
