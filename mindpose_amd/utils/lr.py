"""Learning-rate schedule of the reference recipe (host scalar math):
``WarmupMultiStepDecayLR`` (mindpose/scheduler/warmup_multi_step_decay_lr.py:32-73) - linear warm-up over
``warmup`` steps (mindspore.nn.WarmUpLR: lr * step / warmup), then lr * decay_rate^k where k = number of milestones
(given in epochs) whose first step ``(milestone - 1) * steps_per_epoch`` has been reached.
"""
from typing import Sequence


class WarmupMultiStepDecayLR:
    def __init__(self, lr: float, warmup: int = 0, milestones: Sequence[int] = (), decay_rate: float = 0.1,
                 steps_per_epoch: int = 1) -> None:
        if lr <= 0 or warmup < 0 or steps_per_epoch < 1:
            raise ValueError("bad scheduler arguments")
        self.lr, self.warmup, self.decay_rate = lr, warmup, decay_rate
        self.boundaries = sorted((m - 1) * steps_per_epoch for m in milestones)

    def __call__(self, global_step: int) -> float:
        if global_step < self.warmup:
            return self.lr * min(global_step, self.warmup) / self.warmup
        k = sum(1 for b in self.boundaries if global_step >= b)
        return self.lr * (self.decay_rate ** k)
