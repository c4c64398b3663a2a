"""HRNet head: 1x1 conv C -> K with bias (reference: mindpose/models/heads/hrnet_head.py:14-49)."""
from ...register import register
from ..layers import Conv2d, Plan
from .head import Head


@register("head", extra_name="hrnet_head")
class HRNetHead(Head):
    """``head`` = Conv2d(C -> num_joints, k, bias) with MindSpore's default pad_mode="same" (no padding for k = 1, one pixel for
    k = 3); the parameter names ``head.weight`` / ``head.bias`` are the checkpoint's."""

    def __init__(self, in_channels=32, num_joints=17, final_conv_kernel_size=1):
        super().__init__()
        k = final_conv_kernel_size
        if k not in (1, 3):
            raise ValueError("final_conv_kernel_size must be 1 or 3")
        self.head = Conv2d(in_channels, num_joints, k, padding=k // 2, has_bias=True)

    @property
    def out_joints(self):
        return self.head.out_channels

    def emit(self, plan: Plan, x):
        return plan.conv(x, self.head)

    def train_forward(self, x):
        from .. import train_ops as T
        return T.conv_bn_act(x, self.head, None, relu=False)
