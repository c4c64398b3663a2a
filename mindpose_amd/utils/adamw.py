"""Optimizers of the reference's registry (mindpose/optim/optim_factory.py:9-14) on flat fp32 arenas.

"adamw" - what every recipe uses - is mindspore.nn.AdamWeightDecay = Adam WITHOUT bias correction, eps 1e-6, decoupled
weight decay, and - with ``filter_bias_and_bn`` - no decay for parameters whose name ends in beta / gamma / bias (:17-37).
"adam", "sgd", "momentum", "adagrad" follow the update rules MindSpore documents for those cells.

Natively the parameters, their gradients and the optimizer state are flat fp32 arenas ordered [decay | no-decay], so a step
is two launches of one streaming kernel (``mp_adamw_step`` / ``mp_optimizer_step``) regardless of the number of tensors (878
for HRNet-W32).
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


class _ArenaOptimizer:
    """Parameters, gradients and optimizer state as flat fp32 arenas ordered [decay | no-decay]; ``_update`` of a subclass
    launches its streaming kernel once per group.  ``step`` = gradient mean over ranks (RCCL, overlapped with backward),
    optional dynamic-loss-scale check, update."""

    n_states = 2

    def __init__(self, net: torch.nn.Module, lr: float, weight_decay: float = 0.0, filter_bias_and_bn: bool = True,
                 bucket_mb: float = 32.0, process_group=None, overlap: bool = True, state_init: float = 0.0,
                 transport=None, force_collectives: bool = False) -> None:
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
        self.states = [torch.full_like(self.flat, state_init) for _ in range(self.n_states)]
        self.lr, self.weight_decay = lr, weight_decay
        # gradients: one arena in the SAME order as the parameter arena (no gather before the update), bucketed
        # all-reduce overlapped with backward
        # the 1/world of the gradient mean is folded into the update kernels' gradient scale (mean="consumer"): no extra pass
        self.grads = GradientAverager(self.params, bucket_mb=bucket_mb, process_group=process_group, overlap=overlap,
                                      arena_order="given", mean="consumer", transport=transport,
                                      force_collectives=force_collectives)
        self._grad_flat = self.grads.arena
        self._overflow = torch.zeros(1, device=dev, dtype=torch.int32)  # device flag of the loss-scale overflow check
        self.global_step = 0
        self.time_comm, self.comm_events = False, []

    def zero_grad(self) -> None:
        self.grads.begin_step()

    def close(self) -> None:
        """Release the gradient averager's hooks and communicator (call before ``destroy_process_group``)."""
        self.grads.close()

    def step(self, loss_scale_manager=None) -> bool:
        """All-reduce (mean) the gradients, then one streaming update per decay group.

        With a ``DynamicLossScaleManager`` (amp O2, tools/train.py:170-181) the gradients are first divided by the loss
        scale and checked: on overflow (inf / nan anywhere) the update is skipped and the scale halves, exactly one
        host read-back per step; returns whether the parameters were updated."""
        lib = _lib.load()
        from ..models.train_ops import flush_wgrad_jobs
        flush_wgrad_jobs()  # queued (grouped) weight gradients land in the arena before anything reads it
        if getattr(self, "time_comm", False) and self.grads.active and self._grad_flat.is_cuda:
            # bench.py's DP leg: device time from "all gradients are in the arena" to "the bucket all-reduces have landed" as the
            # compute stream sees it (the native transport runs them on its own stream; finish() makes this stream wait for them)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.grads.finish()
            e1.record()
            self.comm_events.append((e0, e1))
        else:
            self.grads.finish()
        # gradient scale the update kernels apply while reading the arena: 1 / loss_scale (amp O2) x 1 / world (gradient mean)
        self.grad_scale = self.grads.mean_scale
        if loss_scale_manager is not None:
            self.grad_scale /= loss_scale_manager.loss_scale
            # overflow check on the (still scaled) sums: inf / nan survive the scaling, so one read-only pass decides
            self._overflow.zero_()
            _lib.check(lib.mp_grad_finite_check(self._grad_flat.data_ptr(), self._grad_flat.numel(), self._overflow.data_ptr(),
                                                _lib.stream()), "mp_grad_finite_check")
            finite = int(self._overflow.item()) == 0
            loss_scale_manager.update_loss_scale(not finite)
            if not finite:
                return False
        for start, count, wd in ((0, self.n_decay, self.weight_decay), (self.n_decay, self.flat.numel() - self.n_decay, 0.0)):
            if count:
                self._update(lib, start, count, float(wd), _lib.stream())
        from ..models.train_ops import invalidate_packs
        invalidate_packs()  # the kernel wrote the master weights through raw pointers: fp16 packings made before are stale
        self.global_step += 1
        return True


class AdamWeightDecay(_ArenaOptimizer):
    """mindspore.nn.AdamWeightDecay ("adamw"): Adam without bias correction, eps 1e-6, decoupled weight decay."""

    def __init__(self, net: torch.nn.Module, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-6,
                 weight_decay: float = 0.0, **arena_kwargs) -> None:
        super().__init__(net, lr, weight_decay, **arena_kwargs)
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.exp_avg, self.exp_avg_sq = self.states

    def _update(self, lib, start, count, wd, stream):
        _lib.check(lib.mp_adamw_step_scaled(self.flat[start:].data_ptr(), self._grad_flat[start:].data_ptr(),
                                            self.exp_avg[start:].data_ptr(), self.exp_avg_sq[start:].data_ptr(), count,
                                            float(self.lr), float(self.beta1), float(self.beta2), float(self.eps), wd,
                                            float(self.grad_scale), None, stream), "mp_adamw_step_scaled")


class _KernelOptimizer(_ArenaOptimizer):
    """The reference's other registered optimizers (optim_factory.py:9-14) through ``mp_optimizer_step``.  ``loss_scale`` is
    the static scale those MindSpore cells divide the gradient by; weight decay is their L2 term."""

    kind = 0

    def __init__(self, net, lr, weight_decay=0.0, loss_scale: float = 1.0, **arena_kwargs) -> None:
        super().__init__(net, lr, weight_decay, **arena_kwargs)
        self.loss_scale = loss_scale

    def _hyper(self):
        raise NotImplementedError

    def _update(self, lib, start, count, wd, stream):
        import ctypes
        hyper = (ctypes.c_float * 5)(*[float(v) for v in self._hyper()])
        st = [s_[start:].data_ptr() for s_ in self.states] + [None, None]
        _lib.check(lib.mp_optimizer_step(self.kind, self.flat[start:].data_ptr(), self._grad_flat[start:].data_ptr(), st[0], st[1],
                                         count, float(self.lr), float(self.grad_scale) / float(self.loss_scale), wd, ctypes.byref(hyper), stream),
                   "mp_optimizer_step")


class Adam(_KernelOptimizer):
    """mindspore.nn.Adam ("adam", the factory's default name): bias-corrected, eps 1e-8 inside the denominator, L2 decay."""

    kind = 1

    def __init__(self, net, lr: float = 1e-3, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, **kw) -> None:
        super().__init__(net, lr, **kw)
        self.beta1, self.beta2, self.eps = beta1, beta2, eps

    def _hyper(self):
        t = self.global_step + 1
        return self.beta1, self.beta2, self.eps, self.beta1 ** t, self.beta2 ** t


class SGD(_KernelOptimizer):
    """mindspore.nn.SGD ("sgd")."""

    kind, n_states = 2, 1

    def __init__(self, net, lr: float = 0.1, momentum: float = 0.0, dampening: float = 0.0, nesterov: bool = False, **kw) -> None:
        super().__init__(net, lr, **kw)
        self.momentum, self.dampening, self.nesterov = momentum, dampening, nesterov

    def _hyper(self):
        return self.momentum, self.dampening, float(self.nesterov), float(self.global_step == 0), 0.0


class Momentum(_KernelOptimizer):
    """mindspore.nn.Momentum ("momentum")."""

    kind, n_states = 3, 1

    def __init__(self, net, lr: float, momentum: float, use_nesterov: bool = False, **kw) -> None:
        super().__init__(net, lr, **kw)
        self.momentum, self.use_nesterov = momentum, use_nesterov

    def _hyper(self):
        return self.momentum, float(self.use_nesterov), 0.0, 0.0, 0.0


class Adagrad(_KernelOptimizer):
    """mindspore.nn.Adagrad ("adagrad"): accumulator starts at ``accum`` (0.1)."""

    kind, n_states = 4, 1

    def __init__(self, net, lr: float = 1e-3, accum: float = 0.1, **kw) -> None:
        super().__init__(net, lr, state_init=accum, **kw)

    def _hyper(self):
        return 0.0, 0.0, 0.0, 0.0, 0.0
