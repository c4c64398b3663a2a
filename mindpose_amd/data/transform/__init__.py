from .topdown_transform import (TopDownAffine, TopDownBoxToCenterScale, TopDownGenerateTarget,  # noqa: F401
                                get_affine_transform, get_warp_matrix)

__all__ = ["TopDownGenerateTarget", "TopDownBoxToCenterScale", "TopDownAffine", "get_affine_transform", "get_warp_matrix"]
