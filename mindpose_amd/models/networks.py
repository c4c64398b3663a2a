"""Composite networks with the reference's signatures (mindpose/models/networks.py:15-106)."""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .backbones import Backbone
from .decoders import Decoder
from .heads import Head
from .layers import Plan, PlannedModule
from .loss import Loss
from .necks import Neck


class Net(PlannedModule):
    """backbone -> (neck) -> head (networks.py:15-44), recorded as ONE launch plan per input shape."""

    def __init__(self, backbone: Backbone, head: Head, neck: Optional[Neck] = None) -> None:
        super().__init__()
        self.backbone = backbone
        self.head = head
        self.neck = neck
        self.has_neck = self.neck is not None

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        x = self.backbone.emit(plan, x)
        if self.has_neck:
            x = self.neck.emit(plan, x)
        return self.head.emit(plan, x)

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.has_neck:
            raise NotImplementedError("no concrete neck exists in the reference")
        if self.amp_level in ("O2", "O3"):  # fp16 matrix-core training graph; fp32 images in, fp32 heat-maps out
            from . import train_ops as T
            y = self.head.train_forward(self.backbone.train_forward(T.to_c8(x)))
            return T.from_c8(y, self.head.head.out_channels)
        return self.head.train_forward(self.backbone.train_forward(x))


class EvalNet(nn.Module):
    """net + decoder (networks.py:47-76): returns ``(result, raw)`` when ``output_raw``."""

    def __init__(self, net: Net, decoder: Decoder, output_raw: bool = True) -> None:
        super().__init__()
        self.net = net
        self.decoder = decoder
        self.output_raw = output_raw
        self.net.eval()
        self.decoder.eval()

    @torch.no_grad()
    def forward(self, *inputs: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        x = self.net(inputs[0])
        result = self.decoder(x, *inputs[1:])
        if self.output_raw:
            return result, x
        return result


class NetWithLoss(nn.Module):
    """net + loss (networks.py:79-106)."""

    def __init__(self, net: Net, loss: Loss, has_extra_inputs: bool = False) -> None:
        super().__init__()
        self.net = net
        self.loss = loss
        self.has_extra_inputs = has_extra_inputs

    def forward(self, data: torch.Tensor, label: torch.Tensor, *extra_inputs: torch.Tensor) -> torch.Tensor:
        out = self.net(data)
        if self.has_extra_inputs:
            return self.loss(out, label, *extra_inputs)
        return self.loss(out, label)
