"""Seeded synthetic weights for parity tests and the benchmark (no checkpoints / network here).

SURVEY.md 8d: Kaiming-normal conv weights, BN gamma~U(0.5,1.5), beta~N(0,0.1), mean~N(0,0.1),
var~U(0.5,1.5), so BN folding is exercised; the last BN of every residual block gets gamma scaled
by 0.25 (and fuse-row BNs by 0.35) so that activations stay O(1) through ~40 residual blocks and 8 exchange units (otherwise the variance doubles per
block and a 1e-3 heat-map tolerance would be vacuous).  Generated on the CPU generator (device
independent), then copied to the parameters' device.
"""
import math

import torch

from .layers import BatchNorm2d, Conv2d, Conv2dTranspose


def init_synthetic(net: torch.nn.Module, seed: int = 0) -> torch.nn.Module:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    with torch.no_grad():
        for name, m in net.named_modules():
            if isinstance(m, Conv2d):
                fan_in = m.in_channels * m.kernel_size * m.kernel_size
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            elif isinstance(m, Conv2dTranspose):
                # every output pixel sees 2x2 taps x Cin inputs
                fan_in = m.in_channels * 4
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / fan_in))
            elif isinstance(m, BatchNorm2d):
                tail = name.endswith("bn3") or (name.endswith("bn2") and "branches" in name)
                gamma = torch.rand(m.gamma.shape, generator=g) + 0.5
                if tail:
                    gamma = gamma * 0.25
                elif "fuse_layers" in name:
                    gamma = gamma * 0.35  # a fuse row sums up to 4 terms: keep the sum's variance ~ constant
                m.gamma.copy_(gamma)
                m.beta.copy_(torch.randn(m.beta.shape, generator=g) * 0.1)
                m.moving_mean.copy_(torch.randn(m.moving_mean.shape, generator=g) * 0.1)
                m.moving_variance.copy_(torch.rand(m.moving_variance.shape, generator=g) + 0.5)
    if hasattr(net, "invalidate_plans"):
        net.invalidate_plans()
    return net
