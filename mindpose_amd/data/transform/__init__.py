from .topdown_transform import TopDownGenerateTarget  # noqa: F401

__all__ = ["TopDownGenerateTarget"]
