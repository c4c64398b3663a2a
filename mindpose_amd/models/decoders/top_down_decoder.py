"""Top-down heat-map decoder on the MI355X HIP path.

Same constructor, call signature and error behaviour as the reference's ``TopDownHeatMapDecoder``
(mindpose/models/decoders/top_down_decoder.py:14-215): one wavefront per (sample, joint) map does the
arg-max, the +-0.25 shift or the DARK/UDP Taylor refinement (only the 3x3 neighbourhood of the
arg-max is blurred) and the back-projection, in a single kernel.
"""
from typing import Optional, Tuple

import numpy as np
import torch

from ... import _lib
from ...register import register
from .decoder import Decoder


@register("decoder", extra_name="topdown_heatmap")
class TopDownHeatMapDecoder(Decoder):
    def __init__(self, pixel_std: float = 200.0, to_original: bool = True, shift_coordinate: bool = False,
                 use_udp: bool = False, dark_udp_refine: bool = False, kernel_size: int = 11) -> None:
        super().__init__()
        self.pixel_std = pixel_std
        self.to_original = to_original
        self.shift_coordinate = shift_coordinate
        self.use_udp = use_udp
        self.dark_udp_refine = dark_udp_refine
        self.kernel_size = kernel_size
        if self.dark_udp_refine and self.shift_coordinate:
            raise ValueError("`udp_refine` and `shift_coordinate` cannot be `true` in the same time.")
        if self.dark_udp_refine:
            self.register_buffer("gaussian_kernel", self._create_gaussian_kernel(kernel_size), persistent=False)
        else:
            self.gaussian_kernel = None
        self.last_argmax: Optional[torch.Tensor] = None  # [N,K] int32 flat arg-max of the last call

    @property
    def refine_mode(self) -> int:
        if self.shift_coordinate:
            return _lib.MP_REFINE_SHIFT
        if self.dark_udp_refine:
            return _lib.MP_REFINE_DARK
        return _lib.MP_REFINE_NONE

    @staticmethod
    def _create_gaussian_kernel(kernel_size: int) -> torch.Tensor:
        """top_down_decoder.py:207-215 (host numpy, fp64 -> fp32), flattened [k*k]."""
        sigma = 0.3 * ((kernel_size - 1) * 0.5 - 1) + 0.8
        xs = np.arange(-(kernel_size - 1) // 2, (kernel_size - 1) // 2 + 1, 1)
        ys = xs[:, None]
        kernel = np.exp(-(xs ** 2 + ys ** 2) / (2 * sigma ** 2))
        kernel = kernel / kernel.sum()
        return torch.from_numpy(kernel.astype(np.float32).reshape(-1))

    def _blur_on(self, device) -> Optional[torch.Tensor]:
        if self.gaussian_kernel is None:
            return None
        if self.gaussian_kernel.device != device:
            self.gaussian_kernel = self.gaussian_kernel.to(device)
        return self.gaussian_kernel

    def forward(self, heatmap: torch.Tensor, center: torch.Tensor, scale: torch.Tensor,
                score: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """heatmap [N,K,H,W], center/scale [N,2], score [N] -> (preds [N,K,3], boxes [N,6])."""
        lib = _lib.load()
        heatmap = _lib.require_cuda_f32(heatmap, "heatmap")
        dev = heatmap.device
        center = _lib.require_cuda_f32(center.to(dev), "center")
        scale = _lib.require_cuda_f32(scale.to(dev), "scale")
        score = _lib.require_cuda_f32(score.to(dev), "score").reshape(-1)
        n, k, h, w = heatmap.shape
        if center.shape != (n, 2) or scale.shape != (n, 2) or score.shape != (n,):
            raise ValueError("center/scale must be [N,2] and score [N]")
        preds = torch.empty(n, k, 3, device=dev, dtype=torch.float32)
        boxes = torch.empty(n, 6, device=dev, dtype=torch.float32)
        argmax = torch.empty(n, k, device=dev, dtype=torch.int32)
        blur = self._blur_on(dev)
        _lib.check(lib.mp_decode_topdown(
            _lib.ptr(heatmap), _lib.ptr(center), _lib.ptr(scale), _lib.ptr(score), _lib.ptr(preds), _lib.ptr(boxes),
            _lib.ptr(argmax), n, k, h, w, self.refine_mode, int(self.use_udp), int(self.to_original),
            float(self.pixel_std), _lib.ptr(blur), int(self.kernel_size), _lib.stream()), "mp_decode_topdown")
        self.last_argmax = argmax
        return preds, boxes

    def decode_flip_aggregated(self, heatmap: torch.Tensor, flipped_heatmap: torch.Tensor, flip_index: torch.Tensor,
                               shift_heatmap: bool, center: torch.Tensor, scale: torch.Tensor, score: torch.Tensor,
                               return_heatmap: bool = False):
        """Fused ``(h + flip_back(hf)) * 0.5`` + decode (reference: topdown_inferencer.py:165-187 followed by
        the decoder): the averaged heat-map is only written out when ``return_heatmap``."""
        lib = _lib.load()
        heatmap = _lib.require_cuda_f32(heatmap, "heatmap")
        flipped_heatmap = _lib.require_cuda_f32(flipped_heatmap, "flipped_heatmap")
        if heatmap.shape != flipped_heatmap.shape:
            raise ValueError("heatmap and flipped_heatmap must have the same shape")
        dev = heatmap.device
        center = _lib.require_cuda_f32(center.to(dev), "center")
        scale = _lib.require_cuda_f32(scale.to(dev), "scale")
        score = _lib.require_cuda_f32(score.to(dev), "score").reshape(-1)
        flip_index = flip_index.to(dev, torch.int32).contiguous()
        n, k, h, w = heatmap.shape
        if flip_index.numel() != k:
            raise ValueError("flip_index must have K entries")
        preds = torch.empty(n, k, 3, device=dev, dtype=torch.float32)
        boxes = torch.empty(n, 6, device=dev, dtype=torch.float32)
        argmax = torch.empty(n, k, device=dev, dtype=torch.int32)
        avg = torch.empty_like(heatmap) if return_heatmap else None
        blur = self._blur_on(dev)
        _lib.check(lib.mp_flip_aggregate_decode(
            _lib.ptr(heatmap), _lib.ptr(flipped_heatmap), _lib.ptr(flip_index), int(shift_heatmap), _lib.ptr(avg),
            _lib.ptr(center), _lib.ptr(scale), _lib.ptr(score), _lib.ptr(preds), _lib.ptr(boxes), _lib.ptr(argmax),
            n, k, h, w, self.refine_mode, int(self.use_udp), int(self.to_original), float(self.pixel_std),
            _lib.ptr(blur), int(self.kernel_size), _lib.stream()), "mp_flip_aggregate_decode")
        self.last_argmax = argmax
        if return_heatmap:
            return (preds, boxes), avg
        return preds, boxes
