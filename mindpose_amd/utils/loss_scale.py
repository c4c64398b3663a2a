"""Dynamic loss scaling for amp O2 training - ``mindspore.amp.DynamicLossScaleManager`` as tools/train.py:170-173 creates it
(defaults init_loss_scale 2**24, scale_factor 2, scale_window 2000) [MS-knowledge]: the loss is multiplied by the scale
before backward, the optimizer divides the gradients by it; an overflow skips the update and divides the scale by
``scale_factor`` (never below 1), ``scale_window`` consecutive clean steps multiply it."""


class DynamicLossScaleManager:
    def __init__(self, init_loss_scale: float = 2.0 ** 24, scale_factor: float = 2.0, scale_window: int = 2000) -> None:
        if init_loss_scale < 1.0:
            raise ValueError(f"The argument 'init_loss_scale' must be > 1, but got {init_loss_scale}")
        self.loss_scale = float(init_loss_scale)
        self.scale_factor = float(scale_factor)
        self.scale_window = int(scale_window)
        self.cur_iter = 0
        self.last_overflow_iter = -1
        self.skipped_steps = 0

    def get_loss_scale(self) -> float:
        return self.loss_scale

    def scale(self, loss):
        return loss * self.loss_scale

    def update_loss_scale(self, overflow: bool) -> None:
        if overflow:
            self.loss_scale = max(self.loss_scale / self.scale_factor, 1.0)
            self.last_overflow_iter = self.cur_iter
            self.skipped_steps += 1
        elif (self.cur_iter - self.last_overflow_iter) % self.scale_window == 0:
            self.loss_scale *= self.scale_factor
        self.cur_iter += 1
