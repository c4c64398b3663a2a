from .topdown_inferencer import TopDownHeatMapInferencer  # noqa: F401

__all__ = ["TopDownHeatMapInferencer"]
