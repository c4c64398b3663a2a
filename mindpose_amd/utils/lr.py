"""Learning-rate schedules of the reference (host scalar math), callable as ``schedule(global_step) -> lr``.

* ``WarmupMultiStepDecayLR`` (mindpose/scheduler/warmup_multi_step_decay_lr.py:11-77, registry name
  ``warmup_multi_step_decay``): linear warm-up, then the learning rate is multiplied by ``decay_rate`` at the first step of
  every milestone epoch, ``(milestone - 1) * steps_per_epoch`` (the recipe: lr 1e-3, warm-up 500 steps, milestones
  [170, 200] of 210 epochs, configs/hrnet/hrnet_w32_ascend.yaml:75-88).
* ``WarmupCosineDecayLR`` (warmup_cosine_decay_lr.py:12-77, ``warmup_cosine_decay``): linear warm-up, then
  ``mindspore.nn.CosineDecayLR(min_lr, lr, decay_steps)`` of the remaining steps, never below ``min_lr``.
* ``create_lr_scheduler`` (scheduler_factory.py:8-37).

Warm-up is ``mindspore.nn.WarmUpLR``: ``lr * min(step, warmup_steps) / warmup_steps``, used while ``step <= warmup_steps``
(the reference switches on ``global_step > warmup_steps``).  [MS-knowledge for the two mindspore.nn schedules]
"""
import math
from typing import List, Union

from ..register import entrypoint, register


def _warmup_steps(warmup: Union[int, float], total_steps: int) -> int:
    steps = warmup if isinstance(warmup, int) else int(warmup * total_steps)
    if steps > total_steps:
        raise ValueError("Warmup steps must be smaller than total steps")
    return steps


class _WarmupSchedule:
    def __init__(self, lr: float, total_epochs: int, steps_per_epoch: int, warmup: Union[int, float]) -> None:
        self.lr = lr
        self.total_steps = total_epochs * steps_per_epoch
        self.warmup_steps = _warmup_steps(warmup, self.total_steps)

    def _after_warmup(self, global_step: int) -> float:
        raise NotImplementedError

    def step_lr(self, global_step: int) -> float:
        if self.warmup_steps > 0 and not global_step > self.warmup_steps:
            return self.lr * min(global_step, self.warmup_steps) / self.warmup_steps
        return self._after_warmup(global_step)

    __call__ = step_lr


@register("lr_scheduler", extra_name="warmup_multi_step_decay")
class WarmupMultiStepDecayLR(_WarmupSchedule):
    def __init__(self, lr: float, total_epochs: int, steps_per_epoch: int, milestones: List[int], decay_rate: float = 0.1,
                 warmup: Union[int, float] = 0) -> None:
        super().__init__(lr, total_epochs, steps_per_epoch, warmup)
        # (first step, learning rate from that step on): the running product the reference's per-step table holds; like the
        # reference, milestones are consumed in the given order and the last one is the only one that can repeat
        self.breaks = []
        cur, k = lr, 0
        for step in range(self.total_steps):
            if step == (milestones[k] - 1) * steps_per_epoch:
                cur = cur * decay_rate
                k = min(k + 1, len(milestones) - 1)
                self.breaks.append((step, cur))

    def _after_warmup(self, global_step: int) -> float:
        if global_step >= self.total_steps or global_step < 0:
            raise IndexError(f"global_step {global_step} outside the schedule's {self.total_steps} steps")
        lr = self.lr
        for first_step, value in self.breaks:
            if global_step >= first_step:
                lr = value
        return lr


@register("lr_scheduler", extra_name="warmup_cosine_decay")
class WarmupCosineDecayLR(_WarmupSchedule):
    def __init__(self, lr: float, total_epochs: int, steps_per_epoch: int, warmup: Union[int, float] = 0,
                 min_lr: float = 0.0) -> None:
        super().__init__(lr, total_epochs, steps_per_epoch, warmup)
        self.min_lr = min_lr
        self.decay_steps = self.total_steps - self.warmup_steps

    def _after_warmup(self, global_step: int) -> float:
        # mindspore.nn.CosineDecayLR(min_lr, max_lr, decay_steps): the step is clipped to decay_steps
        p = min(global_step - self.warmup_steps, self.decay_steps)
        lr = self.min_lr + 0.5 * (self.lr - self.min_lr) * (1.0 + math.cos(math.pi * p / self.decay_steps))
        return max(lr, self.min_lr)


def create_lr_scheduler(name: str, lr: float, total_epochs: int, steps_per_epoch: int, warmup: Union[int, float] = 0, **kwargs):
    return entrypoint("lr_scheduler", name)(lr=lr, total_epochs=total_epochs, steps_per_epoch=steps_per_epoch, warmup=warmup,
                                            **kwargs)
