"""CPU oracle: JointsMSELoss (TEST INFRASTRUCTURE, see oracle/__init__.py).

numpy restatement of ``JointsMSELoss.construct``
(/root/reference/mindpose/models/loss/mse.py:36-44) + ``nn.MSELoss(reduction="none")`` (:34)
+ ``LossBase.get_loss`` [MS-knowledge: cast fp32, multiply by weights, reduce_mean over ALL axes].
PARITY UNPINNED by the reference (shape-only tests); known answer
``loss(pred=1, target=0, w=0.5) == 0.5`` checked in tests/test_oracle_misc.py.
"""
import numpy as np


def joints_mse(pred, target, target_weight=None, use_target_weight=False):
    """L = mean_{n,k,h,w}( w[n,k] * (pred - target)^2 ); no 0.5 factor, weight applied once,
    mean divides by N*K*H*W regardless of the weights.  Accumulated in float64 -> fp32 so the
    oracle is a stable target for differently-ordered fp32 reductions."""
    pred = np.asarray(pred, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    sq = (pred - target) ** 2
    if use_target_weight:
        w = np.asarray(target_weight, dtype=np.float32)[..., None, None]
        sq = sq * w
    return np.float32(sq.astype(np.float64).mean())


def joints_mse_grad(pred, target, target_weight=None, use_target_weight=False, grad_out=1.0):
    """dL/dpred = grad_out * 2 * w * (pred - target) / (N*K*H*W)."""
    pred = np.asarray(pred, dtype=np.float32)
    target = np.asarray(target, dtype=np.float32)
    g = (pred - target) * np.float32(2.0 * grad_out / pred.size)
    if use_target_weight:
        g = g * np.asarray(target_weight, dtype=np.float32)[..., None, None]
    return g.astype(np.float32)
