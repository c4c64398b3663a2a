from .coco_eval import coco_keypoint_eval  # noqa: F401
from .topdown_evaluator import TopDownEvaluator  # noqa: F401

__all__ = ["TopDownEvaluator", "coco_keypoint_eval"]
