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


_DECONV_PADDING = {4: 1, 2: 0}  # kernel size -> padding that makes Conv2dTranspose(k, stride 2) double H and W exactly


@register("head", extra_name="simple_baseline_head")
class SimpleBaselineHead(Head):
    """Parameter names follow the reference's SequentialCell layout (``deconv_layer.{3i}`` = transposed conv,
    ``deconv_layer.{3i+1}`` = its BatchNorm, ``final_layer`` = the 1x1 conv with bias) so checkpoints map 1:1."""

    def __init__(self, num_deconv_layers: int = 3, num_deconv_filters: List[int] = [256, 256, 256],
                 num_deconv_kernels: List[int] = [4, 4, 4], in_channels: int = 2048, num_joints: int = 17,
                 final_conv_kernel_size: int = 1) -> None:
        super().__init__()
        self.num_deconv_layers, self.in_channels = num_deconv_layers, in_channels
        self.num_deconv_filters, self.num_deconv_kernels = num_deconv_filters, num_deconv_kernels
        cells, width = [], in_channels
        for kernel, planes in list(zip(num_deconv_kernels, num_deconv_filters))[:num_deconv_layers]:
            if kernel not in _DECONV_PADDING:
                raise ValueError("Invalid deconv_kernel.")
            cells.extend((Conv2dTranspose(width, planes, kernel), BatchNorm2d(planes), nn.ReLU()))
            width = planes
        self.deconv_layer = nn.Sequential(*cells)
        self.final_layer = Conv2d(width, num_joints, final_conv_kernel_size, padding=final_conv_kernel_size // 2,
                                  has_bias=True)

    @property
    def out_joints(self) -> int:
        return self.final_layer.out_channels

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        from .. import train_ops as T
        cells = list(self.deconv_layer)
        for deconv, bn in zip(cells[0::3], cells[1::3]):
            x = T.deconv_bn_relu(x, deconv, bn)
        return T.conv_bn_act(x, self.final_layer, None, relu=False)

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        cells = list(self.deconv_layer)
        for deconv, bn in zip(cells[0::3], cells[1::3]):
            x = plan.deconv4x4s2(x, deconv, bn, relu=True)
        return plan.conv(x, self.final_layer)
