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
        # under amp O2 the backbone switches to channel-blocked fp16 itself (HRNet at the image, ResNet after its fp32 stem);
        # fp32 images in, fp32 heat-maps out either way
        from . import train_ops as T
        T.repack_weights(self)  # amp O2: every fp16 weight packing of the step in one launch (no-op in fp32 / on step one)
        y = self.head.train_forward(self.backbone.train_forward(x))
        return T.from_c8(y, self.head.out_joints) if T._is_c8(y) else y


class EvalNet(nn.Module):
    """Inference wrapper with the reference's call contract (networks.py:47-76): ``eval_net(image, *decoder_inputs)`` runs the
    network plan, hands the heat-maps plus the remaining inputs (box centre, scale, score) to the decoder and returns
    ``(decoded, heat-maps)`` - or just ``decoded`` when ``output_raw`` is off.  Both children are put in eval mode."""

    def __init__(self, net: Net, decoder: Decoder, output_raw: bool = True) -> None:
        super().__init__()
        self.net, self.decoder, self.output_raw = net.eval(), decoder.eval(), output_raw

    @torch.no_grad()
    def forward(self, *inputs: torch.Tensor) -> Tuple[torch.Tensor, ...]:
        image, *box_inputs = inputs
        heatmap = self.net(image)
        decoded = self.decoder(heatmap, *box_inputs)
        return (decoded, heatmap) if self.output_raw else decoded


class NetWithLoss(nn.Module):
    """Training wrapper (networks.py:79-106): ``loss(net(data), label[, *extra_inputs])``; the extra inputs (e.g. the
    per-joint target weights of JointsMSELoss) are forwarded only when ``has_extra_inputs`` says the loss takes them."""

    def __init__(self, net: Net, loss: Loss, has_extra_inputs: bool = False) -> None:
        super().__init__()
        self.net, self.loss, self.has_extra_inputs = net, loss, has_extra_inputs

    def forward(self, data: torch.Tensor, label: torch.Tensor, *extra_inputs: torch.Tensor) -> torch.Tensor:
        loss_inputs = (label, *extra_inputs) if self.has_extra_inputs else (label,)
        return self.loss(self.net(data), *loss_inputs)
