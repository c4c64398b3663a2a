"""HRNet head: 1x1 conv C -> K with bias (reference: mindpose/models/heads/hrnet_head.py:14-49)."""
import torch

from ...register import register
from ..layers import Conv2d, Plan
from .head import Head


@register("head", extra_name="hrnet_head")
class HRNetHead(Head):
    def __init__(self, in_channels: int = 32, num_joints: int = 17, final_conv_kernel_size: int = 1) -> None:
        super().__init__()
        if final_conv_kernel_size not in (1, 3):
            raise ValueError("final_conv_kernel_size must be 1 or 3")
        # MindSpore default pad_mode="same": no padding for k=1, 1 for k=3
        self.head = Conv2d(in_channels, num_joints, final_conv_kernel_size,
                           padding=final_conv_kernel_size // 2, has_bias=True)

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        return plan.conv(x, self.head)

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        from .. import train_ops as T
        return T.conv_bn_act(x, self.head, None, relu=False)
