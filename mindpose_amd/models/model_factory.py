"""Factories with the reference's names and signatures (mindpose/models/model_factory.py:24-203)."""
from typing import Any, Dict, Optional

from ..register import entrypoint
from .backbones import Backbone
from .decoders import Decoder
from .heads import Head
from .loss import Loss
from .necks import Neck
from .networks import EvalNet, Net, NetWithLoss

__all__ = [
    "create_backbone",
    "create_neck",
    "create_head",
    "create_decoder",
    "create_loss",
    "create_network",
    "create_eval_network",
    "create_network_with_loss",
]


def create_backbone(name: str, pretrained: bool = False, ckpt_url: str = "", in_channels: int = 3,
                    **kwargs: Any) -> Backbone:
    return entrypoint("backbone", name)(pretrained=pretrained, ckpt_url=ckpt_url, in_channels=in_channels, **kwargs)


def create_head(name: str, in_channels, num_joints: int = 17, **kwargs: Any) -> Head:
    return entrypoint("head", name)(in_channels=in_channels, num_joints=num_joints, **kwargs)


def create_neck(name: str, in_channels, out_channels, **kwargs: Any) -> Neck:
    return entrypoint("neck", name)(in_channels=in_channels, out_channels=out_channels, **kwargs)


def create_decoder(name: str, **kwargs: Any) -> Decoder:
    return entrypoint("decoder", name)(**kwargs)


def create_loss(name: str, **kwargs: Any) -> Loss:
    return entrypoint("loss", name)(**kwargs)


def create_network(backbone_name: str, head_name: str, neck_name: str = "", backbone_pretrained: bool = False,
                   backbone_ckpt_url: str = "", in_channels: int = 3, neck_out_channels: int = 256,
                   num_joints: int = 17, backbone_args: Optional[Dict[str, Any]] = None,
                   neck_args: Optional[Dict[str, Any]] = None, head_args: Optional[Dict[str, Any]] = None) -> Net:
    backbone_args = backbone_args if backbone_args else dict()
    neck_args = neck_args if neck_args else dict()
    head_args = head_args if head_args else dict()
    backbone = create_backbone(backbone_name, pretrained=backbone_pretrained, ckpt_url=backbone_ckpt_url,
                               in_channels=in_channels, **backbone_args)
    if neck_name:
        neck = create_neck(neck_name, in_channels=backbone.out_channels, out_channels=neck_out_channels, **neck_args)
        head = create_head(head_name, in_channels=neck.out_channels, num_joints=num_joints, **head_args)
    else:
        neck = None
        head = create_head(head_name, in_channels=backbone.out_channels, num_joints=num_joints, **head_args)
    return Net(backbone, head, neck=neck)


def create_eval_network(net: Net, decoder: Decoder, output_raw: bool = True) -> EvalNet:
    return EvalNet(net, decoder, output_raw=output_raw)


def create_network_with_loss(net: Net, loss: Loss, has_extra_inputs: bool = False) -> NetWithLoss:
    return NetWithLoss(net, loss, has_extra_inputs=has_extra_inputs)
