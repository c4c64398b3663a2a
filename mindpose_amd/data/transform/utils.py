"""Geometry helpers with the reference's names (mindpose/data/transform/utils.py), numpy on the host like the reference.

``fliplr_joints`` / ``get_affine_transform`` / ``get_warp_matrix`` live next to the transforms that use them
(``topdown_transform.py``) and are re-exported here; the point helpers below complete the module's surface.  All of them are
pinned to the reference's own outputs (``tests/golden/helpers.npz``, ``geometry.npz``).
"""
from typing import List, Sequence

import numpy as np

from .topdown_transform import fliplr_joints, get_affine_transform, get_warp_matrix  # noqa: F401

__all__ = ["fliplr_joints", "get_affine_transform", "get_warp_matrix", "affine_transform", "rotate_point", "warp_affine_joints",
           "pad_to_same", "transform_keypoints"]


def affine_transform(pt: Sequence[float], trans_mat: np.ndarray) -> np.ndarray:
    """One 2-D point through a 2x3 affine matrix (utils.py:101-114)."""
    if len(pt) != 2:
        raise AssertionError("affine_transform takes one 2-D point")
    return np.asarray(trans_mat) @ np.array([pt[0], pt[1], 1.0])


def rotate_point(pt: Sequence[float], angle_rad: float) -> List[float]:
    """Rotate a 2-D point about the origin (utils.py:117-133); returns a list, like the reference."""
    if len(pt) != 2:
        raise AssertionError("rotate_point takes one 2-D point")
    s, c = np.sin(angle_rad), np.cos(angle_rad)
    return [pt[0] * c - pt[1] * s, pt[0] * s + pt[1] * c]


def warp_affine_joints(joints: np.ndarray, mat: np.ndarray) -> np.ndarray:
    """[..., 2] joint coordinates through a 2x3 affine matrix (utils.py:193-210): homogeneous coordinate appended as fp32."""
    homogeneous = np.concatenate([joints, np.ones(joints.shape[:-1] + (1,), dtype=np.float32)], axis=-1)
    return homogeneous @ mat.T


def pad_to_same(arrays: List[np.ndarray]) -> List[np.ndarray]:
    """Zero-pad (at the end of every axis) to the element-wise maximum shape (utils.py:213-232)."""
    target = np.max([a.shape for a in arrays], axis=0)
    return [np.pad(a, [(0, int(t - s)) for s, t in zip(a.shape, target)]) for a in arrays]


def transform_keypoints(coords: List[np.ndarray], center: np.ndarray, scale: np.ndarray, heatmap_shape: np.ndarray,
                        pixel_std: float = 200.0) -> List[np.ndarray]:
    """Heat-map coordinates -> original image, per image of a batch (utils.py:235-275): x' = x * s_x / W + c_x - s_x / 2 with
    s = scale * pixel_std; empty detections pass through."""
    size = scale * pixel_std
    per_pixel = size / heatmap_shape[:, :2]
    out = []
    for i, c in enumerate(coords):
        if c.size == 0:
            out.append(c)
            continue
        t = c.copy()
        for axis in (0, 1):
            t[:, :, axis] = c[:, :, axis] * per_pixel[i, axis] + center[i, axis] - size[i, axis] * 0.5
        out.append(t)
    return out
