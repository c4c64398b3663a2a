"""Factories with the reference's names and call signatures (mindpose/models/model_factory.py:24-203).

The surface is the drop-in contract (``create_network("hrnet_w32", "hrnet_head", ...)`` etc. must keep working when the
import changes from ``mindpose`` to ``mindpose_amd``); the implementation is one registry lookup helper plus the wiring of
backbone -> optional neck -> head, whose input width is whatever the previous stage reports as ``out_channels``.
"""
from ..register import entrypoint
from .networks import EvalNet, Net, NetWithLoss

__all__ = ["create_backbone", "create_neck", "create_head", "create_decoder", "create_loss", "create_network",
           "create_eval_network", "create_network_with_loss"]

# Argument types (as in the reference): name -> str registry key; in_channels / out_channels / num_joints -> int;
# pretrained / backbone_pretrained / output_raw / has_extra_inputs -> bool; ckpt_url -> str; *_args -> optional dict of
# constructor keywords; the create_* functions return Backbone / Head / Neck / Decoder / Loss / Net / EvalNet / NetWithLoss.


def _instantiate(kind, name, **ctor_kwargs):
    """Resolve ``name`` in registry section ``kind`` (ValueError listing the alternatives on a miss) and construct it."""
    return entrypoint(kind, name)(**ctor_kwargs)


def create_backbone(name, pretrained=False, ckpt_url="", in_channels=3, **kwargs):
    return _instantiate("backbone", name, pretrained=pretrained, ckpt_url=ckpt_url, in_channels=in_channels, **kwargs)


def create_head(name, in_channels, num_joints=17, **kwargs):
    return _instantiate("head", name, in_channels=in_channels, num_joints=num_joints, **kwargs)


def create_neck(name, in_channels, out_channels, **kwargs):
    return _instantiate("neck", name, in_channels=in_channels, out_channels=out_channels, **kwargs)


def create_decoder(name, **kwargs):
    return _instantiate("decoder", name, **kwargs)


def create_loss(name, **kwargs):
    return _instantiate("loss", name, **kwargs)


def create_network(backbone_name, head_name, neck_name="", backbone_pretrained=False, backbone_ckpt_url="", in_channels=3,
                   neck_out_channels=256, num_joints=17, backbone_args=None, neck_args=None, head_args=None):
    stages = [create_backbone(backbone_name, backbone_pretrained, backbone_ckpt_url, in_channels, **(backbone_args or {}))]
    if neck_name:
        stages.append(create_neck(neck_name, stages[-1].out_channels, neck_out_channels, **(neck_args or {})))
    head = create_head(head_name, stages[-1].out_channels, num_joints, **(head_args or {}))
    return Net(stages[0], head, neck=stages[1] if neck_name else None)


def create_eval_network(net, decoder, output_raw=True):
    return EvalNet(net, decoder, output_raw=output_raw)


def create_network_with_loss(net, loss, has_extra_inputs=False):
    return NetWithLoss(net, loss, has_extra_inputs=has_extra_inputs)
