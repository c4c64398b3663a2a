"""``mindpose.optim`` surface: ``create_optimizer`` (mindpose/optim/optim_factory.py)."""
from ..utils.optim_factory import create_optimizer  # noqa: F401

__all__ = ["create_optimizer"]
