"""JointsMSELoss on the MI355X HIP path (reference: mindpose/models/loss/mse.py:11-44).

L = mean_{n,k,h,w}( w[n,k] * (pred - target)^2 ): one pass over pred/target (16 B per lane), a
deterministic two-stage reduction, and an analytic backward kernel wired through autograd.
"""
from typing import Optional

import torch

from ... import _lib
from ...register import register
from .loss import Loss


class _JointsMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight):
        lib = _lib.load()
        n, k, h, w = pred.shape
        ws_bytes = lib.mp_joints_mse_workspace_bytes(n, k)
        ws = torch.empty(ws_bytes // 4, device=pred.device, dtype=torch.float32)
        loss = torch.empty(1, device=pred.device, dtype=torch.float32)
        _lib.check(lib.mp_joints_mse_fwd(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight), _lib.ptr(loss),
                                         _lib.ptr(ws), ws_bytes, n, k, h * w, _lib.stream()), "mp_joints_mse_fwd")
        ctx.save_for_backward(pred, target, weight if weight is not None else torch.empty(0, device=pred.device))
        ctx.has_weight = weight is not None
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        pred, target, weight = ctx.saved_tensors
        n, k, h, w = pred.shape
        grad = torch.empty_like(pred)
        go = grad_out.detach().float().reshape(1).contiguous()
        _lib.check(lib.mp_joints_mse_bwd(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weight) if ctx.has_weight else None,
                                         _lib.ptr(go), _lib.ptr(grad), n, k, h * w, _lib.stream()), "mp_joints_mse_bwd")
        return grad, None, None


@register("loss", extra_name="joint_mse")
class JointsMSELoss(Loss):
    def __init__(self, use_target_weight: bool = False, reduction: Optional[str] = "mean") -> None:
        super().__init__(reduction=reduction)
        if reduction != "mean":
            raise NotImplementedError("only reduction='mean' (the reference recipes' setting) runs on the HIP path")
        self.use_target_weight = use_target_weight

    def forward(self, pred: torch.Tensor, target: torch.Tensor, target_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
        pred = _lib.require_cuda_f32(pred, "pred")
        target = _lib.require_cuda_f32(target, "target")
        if pred.shape != target.shape or pred.dim() != 4:
            raise ValueError("pred and target must both be [N,K,H,W]")
        weight = None
        if self.use_target_weight:
            if target_weight is None:
                raise ValueError("target_weight is required when use_target_weight=True")
            weight = _lib.require_cuda_f32(target_weight, "target_weight").reshape(pred.shape[0], pred.shape[1])
        return _JointsMSEFn.apply(pred, target, weight)
