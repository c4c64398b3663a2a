"""``create_optimizer`` with the reference's signature (mindpose/optim/optim_factory.py:40-72).

Every name the reference registers resolves: "adamw" (``mindspore.nn.AdamWeightDecay``, what all recipes use; ``utils/adamw.py``
-> ``mp_adamw_step``) and "adam" (the factory's default), "sgd", "momentum", "adagrad" (-> ``mp_optimizer_step``; update rules
from the MindSpore documentation, not pinned by any reference test).  ``params`` is a module (its named parameters are
grouped) or an iterable of ``(name, parameter)`` pairs; as in the reference, weight decay is applied only through the decay /
no-decay grouping, i.e. when ``weight_decay`` is non-zero AND ``filter_bias_and_bn`` is set (parameters whose name contains
beta / gamma / bias are not decayed); ``learning_rate`` may be a float or a schedule ``f(global_step) -> lr`` that ``step()``
consults; ``loss_scale`` (static) goes to every optimizer but "adamw", as in the reference (:64-72).
"""
from typing import Any, Callable, Union

import torch

from .adamw import Adagrad, Adam, AdamWeightDecay, Momentum, SGD


class _NamedParams(torch.nn.Module):
    """Adapter: an iterable of (name, parameter) pairs presented as a module for AdamWeightDecay's flat arenas."""

    def __init__(self, named) -> None:
        super().__init__()
        self._named = list(named)

    def named_parameters(self, *args, **kwargs):
        return iter(self._named)


_OPTIMIZERS = {"adamw": AdamWeightDecay, "AdamWeightDecay": AdamWeightDecay, "adam": Adam, "Adam": Adam, "sgd": SGD, "SGD": SGD,
               "momentum": Momentum, "Momentum": Momentum, "adagrad": Adagrad, "Adagrad": Adagrad}


def _scheduled(base):
    class Scheduled(base):
        def __init__(self, net, learning_rate: Union[float, Callable[[int], float]], **kwargs: Any) -> None:
            self._schedule = learning_rate if callable(learning_rate) else None
            super().__init__(net, lr=learning_rate(0) if callable(learning_rate) else learning_rate, **kwargs)

        def step(self, loss_scale_manager=None) -> bool:
            if self._schedule is not None:
                self.lr = float(self._schedule(self.global_step))
            return super().step(loss_scale_manager=loss_scale_manager)

    Scheduled.__name__ = base.__name__
    return Scheduled


def create_optimizer(params, name: str = "adam", learning_rate=0.001, weight_decay: float = 0.0, filter_bias_and_bn: bool = True,
                     loss_scale: float = 1.0, **kwargs: Any):
    if name not in _OPTIMIZERS:
        raise ValueError(f"Unkown components `{name}`. Supported componetns in `optim`: {sorted(_OPTIMIZERS)}")
    net = params if isinstance(params, torch.nn.Module) else _NamedParams(params)
    grouped = bool(weight_decay) and filter_bias_and_bn
    if _OPTIMIZERS[name] is not AdamWeightDecay:
        kwargs["loss_scale"] = loss_scale
    return _scheduled(_OPTIMIZERS[name])(net, learning_rate, weight_decay=weight_decay if grouped else 0.0, filter_bias_and_bn=True,
                                         **kwargs)
