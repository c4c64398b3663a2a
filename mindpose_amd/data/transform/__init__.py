from .topdown_transform import (TopDownAffine, TopDownBoxToCenterScale, TopDownGenerateTarget,  # noqa: F401
                                TopDownHalfBodyTransform, TopDownHorizontalRandomFlip, TopDownRandomScaleRotation,
                                fliplr_joints, get_affine_transform, get_warp_matrix)

__all__ = ["TopDownGenerateTarget", "TopDownBoxToCenterScale", "TopDownAffine", "TopDownHorizontalRandomFlip",
           "TopDownHalfBodyTransform", "TopDownRandomScaleRotation", "fliplr_joints", "get_affine_transform", "get_warp_matrix"]
