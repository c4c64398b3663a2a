"""Gaussian heat-map target generation on the MI355X HIP path.

Host mirror of the reference's ``TopDownGenerateTarget``
(mindpose/data/transform/topdown_transform.py:264-430): same constructor / config keys / error
behaviour, but BATCHED and on the device - keypoints [N,K,3] in, target [N,K,H,W] + weights [N,K] out,
written once by one block per (sample, joint) - instead of a 17-iteration Python loop per sample in
dataset worker processes.  The loader geometry transforms around it are "next" rows.
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from ... import _lib
from ...register import register


@register("transform", extra_name="topdown_generate_target")
class TopDownGenerateTarget:
    def __init__(self, is_train: bool = True, config: Optional[Dict[str, Any]] = None, sigma: float = 2.0,
                 use_different_joint_weights: bool = False, use_udp: bool = False) -> None:
        self.is_train = is_train
        self.config = config if config else dict()
        self._transform_cfg = self.load_transform_cfg()
        self.sigma = sigma
        self.use_different_joint_weights = use_different_joint_weights
        self.use_udp = use_udp
        if self.use_different_joint_weights and self._transform_cfg["joint_weights"] is None:
            raise ValueError("`joint_weights` must be provided if `use_different_joint_weights` is True.")
        self._patch_cache: Dict[str, torch.Tensor] = {}
        self._jw_cache: Dict[str, torch.Tensor] = {}

    def load_transform_cfg(self) -> Dict[str, Any]:
        """Subset of topdown_transform.py:60-94 that target generation reads."""
        cfg = dict()
        cfg["image_size"] = np.array(self.config["image_size"])
        cfg["heatmap_size"] = np.array(self.config["heatmap_size"])
        assert len(cfg["image_size"]) == 2
        assert len(cfg["heatmap_size"]) == 2
        cfg["joint_weights"] = np.array(self.config["joint_weights"]) if "joint_weights" in self.config else None
        return cfg

    # -- host-side constants computed with the reference's own numpy expressions ------------------
    def _feat_stride(self) -> Tuple[float, float]:
        image_size = self._transform_cfg["image_size"]
        w, h = self._transform_cfg["heatmap_size"]
        if self.use_udp:
            fs = (image_size - 1.0) / (np.array([w, h]) - 1.0)  # :398
        else:
            fs = image_size / np.array([w, h])  # :349
        return float(fs[0]), float(fs[1])

    def _gaussian_patch(self) -> np.ndarray:
        """:335-344 - the un-normalised patch, bit-identical to the reference on the same numpy."""
        tmp_size = self.sigma * 3
        size = 2 * tmp_size + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, None]
        x0, y0 = size // 2, size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * self.sigma ** 2))
        return np.ascontiguousarray(g, dtype=np.float32)

    def generate(self, keypoints: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """keypoints [N,K,3] (x, y, visibility) in input-image px, CUDA -> (target [N,K,H,W], weight [N,K])."""
        lib = _lib.load()
        kp = _lib.require_cuda_f32(keypoints, "keypoints")
        if kp.dim() != 3 or kp.shape[2] != 3:
            raise ValueError("keypoints must be [N,K,3]")
        n, k, _ = kp.shape
        w, h = (int(v) for v in self._transform_cfg["heatmap_size"])
        dev = kp.device
        key = str(dev)
        patch, side = None, 0
        if not self.use_udp:
            if key not in self._patch_cache:
                self._patch_cache[key] = torch.from_numpy(self._gaussian_patch()).to(dev)
            patch = self._patch_cache[key]
            side = patch.shape[0]
        jw = None
        if self.use_different_joint_weights:
            if key not in self._jw_cache:
                jwn = np.asarray(self._transform_cfg["joint_weights"], dtype=np.float64).reshape(-1)
                if jwn.size != k:
                    raise ValueError("joint_weights must have K entries")
                self._jw_cache[key] = torch.from_numpy(jwn).to(dev)
            jw = self._jw_cache[key]
        fsx, fsy = self._feat_stride()
        target = torch.empty(n, k, h, w, device=dev, dtype=torch.float32)
        weight = torch.empty(n, k, device=dev, dtype=torch.float32)
        _lib.check(lib.mp_gaussian_target(_lib.ptr(kp), _lib.ptr(patch), side, _lib.ptr(jw), _lib.ptr(target),
                                          _lib.ptr(weight), n, k, h, w, fsx, fsy, float(self.sigma),
                                          int(self.use_udp), _lib.stream()), "mp_gaussian_target")
        return target, weight

    def transform(self, state: Dict[str, Any]) -> Dict[str, Any]:
        """Reference-shaped entry (:305-322): ``state["keypoints"]`` [K,3] (numpy or tensor) ->
        ``dict(target [K,H,W], target_weight [K])`` as device tensors."""
        kp = state["keypoints"]
        if not torch.is_tensor(kp):
            kp = torch.as_tensor(np.asarray(kp, dtype=np.float32)).cuda()
        target, weight = self.generate(kp[None])
        return dict(target=target[0], target_weight=weight[0])

    def __call__(self, keypoints: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.generate(keypoints)
