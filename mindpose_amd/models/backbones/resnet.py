"""ResNet-50/101/152 backbones on the MI355X HIP path (reference: mindpose/models/backbones/resnet.py).

Same graph / registry names / parameter names; stride sits on the 3x3 conv of the bottleneck
(resnet.py:74-109), the stem is 7x7 s2 p3 + BN + ReLU + MaxPool(3, 2, "same") (:180-190).
"""
from typing import List, Type

import torch
import torch.nn as nn

from ...register import register
from ..layers import BatchNorm2d, Conv2d, Plan
from .backbone import Backbone
from .hrnet import Bottleneck as _HRBottleneck
from .hrnet import _conv_bn, emit_block_sequence
from .utils import load_pretrained

__all__ = ["ResNet", "resnet50", "resnet101", "resnet152"]


class Bottleneck(_HRBottleneck):
    """resnet.py:74-138 - identical arithmetic to the HRNet bottleneck (stride on conv2)."""


@register("backbone")
class ResNet(Backbone):
    """ResNet backbone - resnet.py:142-273."""

    def __init__(self, block: Type[Bottleneck], layers: List[int], in_channels: int = 3) -> None:
        super().__init__()
        self.input_channels = 64
        self.conv1 = Conv2d(in_channels, 64, 7, stride=2, padding=3)
        self.bn1 = BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)

    def _make_layer(self, block, channels: int, block_nums: int, stride: int = 1) -> nn.Sequential:
        down = None
        if stride != 1 or self.input_channels != channels * block.expansion:
            down = _conv_bn(self.input_channels, channels * block.expansion, 1, stride=stride)
        layers = [block(self.input_channels, channels, stride=stride, down_sample=down)]
        self.input_channels = channels * block.expansion
        for _ in range(1, block_nums):
            layers.append(block(self.input_channels, channels))
        return nn.Sequential(*layers)

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        """Recorded form of ``forward_feature`` resnet.py:247-264."""
        x = plan.conv(x, self.conv1, self.bn1, relu=True)  # 7x7 stem + max-pool stay fp32 NCHW under amp O2 too
        x = plan.maxpool3x3s2_same(x)
        x = plan.enter(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            # (layer1 has HRNet stage 1's widths - 64 -> 256 Bottlenecks on the 64x48 map: its expand / reduce / down-sample 1x1 convs
            # take the chain launches where the plan has them)
            x = emit_block_sequence(plan, list(layer), x)
        return x

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """Training form of ``forward_feature`` resnet.py:247-264: fp32 stem + max-pool, then the bottleneck stages in the
        network's precision (channel-blocked fp16 under amp O2)."""
        from .. import train_ops as T
        x = T.maxpool3x3s2_same(T.stem_conv_bn_relu(x, self.conv1, self.bn1))
        if self.amp_level in ("O2", "O3"):
            x = T.to_c8(x)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk.train_forward(x)
        return x

    @property
    def out_channels(self) -> int:
        return 512 * Bottleneck.expansion


def _resnet(layers, pretrained, ckpt_url, in_channels, **kwargs) -> ResNet:
    model = ResNet(Bottleneck, layers, in_channels=in_channels, **kwargs)
    if pretrained:
        load_pretrained(model, ckpt_url=ckpt_url)
    return model


@register("backbone")
def resnet50(pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3, **kwargs) -> ResNet:
    """resnet.py:277-295."""
    return _resnet([3, 4, 6, 3], pretrained, ckpt_url, in_channels, **kwargs)


@register("backbone")
def resnet101(pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3, **kwargs) -> ResNet:
    """resnet.py:299-317."""
    return _resnet([3, 4, 23, 3], pretrained, ckpt_url, in_channels, **kwargs)


@register("backbone")
def resnet152(pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3, **kwargs) -> ResNet:
    """resnet.py:321-339."""
    return _resnet([3, 8, 36, 3], pretrained, ckpt_url, in_channels, **kwargs)
