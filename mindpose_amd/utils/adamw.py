"""AdamWeightDecay as the reference configures it for "adamw" (mindpose/optim/optim_factory.py:10-72):
mindspore.nn.AdamWeightDecay = Adam WITHOUT bias correction, eps 1e-6, decoupled weight decay, and - with
``filter_bias_and_bn`` - no decay for parameters whose name ends in beta / gamma / bias (:17-37).

Natively the parameters, their gradients and both moments are flat fp32 arenas ordered [decay | no-decay], so a step is
two launches of one streaming kernel (``mp_adamw_step``) regardless of the number of tensors (878 for HRNet-W32).
"""
from typing import Iterable, Tuple

import torch

from .. import _lib
from .grad_allreduce import GradientAverager


def split_decay(named_params: Iterable[Tuple[str, torch.nn.Parameter]], filter_bias_and_bn: bool = True):
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        if filter_bias_and_bn and name.endswith(("beta", "gamma", "bias")):
            no_decay.append(p)
        else:
            decay.append(p)
    return decay, no_decay


class AdamWeightDecay:
    def __init__(self, net: torch.nn.Module, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-6,
                 weight_decay: float = 0.0, filter_bias_and_bn: bool = True, bucket_mb: float = 32.0, process_group=None,
                 overlap: bool = True) -> None:
        decay, no_decay = split_decay(net.named_parameters(), filter_bias_and_bn)
        self.params = decay + no_decay
        self.n_decay = sum(p.numel() for p in decay)
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        # flat parameter arena: parameters become views (same values, same names)
        self.flat = torch.empty(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.lr, self.beta1, self.beta2, self.eps, self.weight_decay = lr, beta1, beta2, eps, weight_decay
        # gradients: one arena in the SAME order as the parameter arena (no gather before the update), bucketed
        # all-reduce overlapped with backward
        self.grads = GradientAverager(self.params, bucket_mb=bucket_mb, process_group=process_group, overlap=overlap,
                                      arena_order="given")
        self._grad_flat = self.grads.arena
        self.global_step = 0

    def zero_grad(self) -> None:
        self.grads.begin_step()

    def step(self, loss_scale_manager=None) -> bool:
        """All-reduce (mean) the gradients, then one fused update per decay group.

        With a ``DynamicLossScaleManager`` (amp O2, tools/train.py:170-181) the gradients are first divided by the loss
        scale and checked: on overflow (inf / nan anywhere) the update is skipped and the scale halves, exactly one
        host read-back per step; returns whether the parameters were updated."""
        lib = _lib.load()
        self.grads.finish()
        if loss_scale_manager is not None:
            self._grad_flat.mul_(1.0 / loss_scale_manager.loss_scale)
            finite = bool(torch.isfinite(self._grad_flat).all())
            loss_scale_manager.update_loss_scale(not finite)
            if not finite:
                return False
        s = _lib.stream()
        for start, count, wd in ((0, self.n_decay, self.weight_decay), (self.n_decay, self.flat.numel() - self.n_decay, 0.0)):
            if count == 0:
                continue
            _lib.check(lib.mp_adamw_step(self.flat[start:].data_ptr(), self._grad_flat[start:].data_ptr(),
                                         self.exp_avg[start:].data_ptr(), self.exp_avg_sq[start:].data_ptr(), count,
                                         float(self.lr), float(self.beta1), float(self.beta2), float(self.eps), float(wd), s),
                       "mp_adamw_step")
        from ..models.train_ops import invalidate_packs
        invalidate_packs()  # the kernel wrote the master weights through raw pointers: fp16 packings made before are stale
        self.global_step += 1
        return True
