"""SimpleBaseline head (reference: mindpose/models/heads/simple_baseline_head.py:17-98).

3 x [Conv2dTranspose(k4,s2,p1) + BN + ReLU] then a 1x1 conv with bias.  Each transposed conv runs as
four 2x2 sub-pixel phase convolutions on the un-dilated input (no zero insertion), BN+ReLU fused.
"""
from typing import List

import torch
import torch.nn as nn

from ...register import register
from ..layers import BatchNorm2d, Conv2d, Conv2dTranspose, Plan
from .head import Head


@register("head", extra_name="simple_baseline_head")
class SimpleBaselineHead(Head):
    def __init__(self, num_deconv_layers: int = 3, num_deconv_filters: List[int] = [256, 256, 256],
                 num_deconv_kernels: List[int] = [4, 4, 4], in_channels: int = 2048, num_joints: int = 17,
                 final_conv_kernel_size: int = 1) -> None:
        super().__init__()
        self.num_deconv_layers = num_deconv_layers
        self.num_deconv_filters = num_deconv_filters
        self.num_deconv_kernels = num_deconv_kernels
        self.in_channels = in_channels
        self.deconv_layer = self.make_deconv_layer()
        self.final_layer = Conv2d(num_deconv_filters[-1], num_joints, final_conv_kernel_size,
                                  padding=final_conv_kernel_size // 2, has_bias=True)

    def _get_deconv_padding(self, deconv_kernel: int) -> int:
        if deconv_kernel == 4:
            return 1
        if deconv_kernel == 2:
            return 0
        raise ValueError("Invalid deconv_kernel.")

    def make_deconv_layer(self) -> nn.Sequential:
        layers = []
        cin = self.in_channels
        for i in range(self.num_deconv_layers):
            self._get_deconv_padding(self.num_deconv_kernels[i])
            planes = self.num_deconv_filters[i]
            layers += [Conv2dTranspose(cin, planes, self.num_deconv_kernels[i]), BatchNorm2d(planes), nn.ReLU()]
            cin = planes
        return nn.Sequential(*layers)

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        for i in range(self.num_deconv_layers):
            x = plan.deconv4x4s2(x, self.deconv_layer[3 * i], self.deconv_layer[3 * i + 1], relu=True)
        return plan.conv(x, self.final_layer)
