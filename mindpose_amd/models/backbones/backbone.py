from typing import List, Union

import torch

from ..layers import Plan, PlannedModule


class Backbone(PlannedModule):
    """Abstract class for all backbones (reference: mindpose/models/backbones/backbone.py:7-36).

    Child classes implement ``emit`` (the recorded form of the reference's ``forward_feature``) and
    ``out_channels``.
    """

    def forward_feature(self, x: torch.Tensor) -> torch.Tensor:
        """Perform the feature extraction (reference name kept; runs the HIP launch plan)."""
        return PlannedModule.forward(self, x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_feature(x)

    def emit(self, plan: Plan, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("Child class must implement this method.")

    @property
    def out_channels(self) -> Union[List[int], int]:
        raise NotImplementedError("Child class must implement this method.")
