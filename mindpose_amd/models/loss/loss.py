import torch.nn as nn


class Loss(nn.Module):
    """Abstract class for all losses (reference: mindpose/models/loss/loss.py:4)."""

    def __init__(self, reduction: str = "mean") -> None:
        super().__init__()
        if reduction not in ("mean", "sum", "none", None):
            raise ValueError(f"reduction method for {reduction} is not supported")
        self.reduction = reduction
